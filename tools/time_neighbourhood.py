#!/usr/bin/env python3
"""Wall time of the per-step neighbourhood work of the bench stack (ball query + source-major edge list per level),
per level and per phase.  Run under `rocprofv3 --kernel-trace --stats` for the kernel view."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W
import bench

dev = torch.device("cuda:0")
levels = W.build_stack(W.WORKLOADS['headline'], dev, 0)

def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

tot = 0.0
for i, lv in enumerate(levels):
    pc, r = lv["pc"], lv["r"]
    bq = timed(lambda: amd.ops.ball_query(pc.pts_, pc.pts_, pc.batch_ids_, pc.batch_ids_, r, pc.num_batches()))
    nb, _ = amd.ops.ball_query(pc.pts_, pc.pts_, pc.batch_ids_, pc.batch_ids_, r)
    tr = timed(lambda: amd.ops.csr_transpose(nb, lv["n"]))

    def layer_view():  # what a conv layer triggers on first use of a fresh neighbourhood
        amd.layers._geometry_of(pc, pc, amd.pc.BQNeighborhood(pc, pc, r)).transpose()

    full = timed(layer_view)
    tot += full
    print(f"level {i}: n {lv['n']:6d} e {lv['e']:8d}  ball_query {bq:.3f} ms   BQNeighborhood + geometry views {full:.3f} ms"
          f"   (explicit csr_transpose, not needed for a cloud against itself: {tr:.3f} ms)")
print(f"total {tot:.3f} ms")
