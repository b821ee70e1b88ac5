#!/usr/bin/env python3
"""Per-stage times (library HIP events, eager launches) of every level of the bench stack, and the graph-replayed
time of each level alone: where the small levels spend their time."""
import os, sys, time
import ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import se3conv3d_amd as amd
from se3conv3d_amd import _lib
import bench

dev = torch.device("cuda:0")
lib = _lib.load()
from se3conv3d_amd import workloads as W
amd.set_precision(os.environ.get("SE3CONV_PRECISION", "bf16x3"))  # fp32: the exact mode's levels
levels = W.build_stack(W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'headline'], dev, 0)
for i, lv in enumerate(levels):
    for _ in range(3):
        bench.step([lv])          # warm-up: lazy builds, allocator, clocks
    torch.cuda.synchronize()
    st = bench.profile_level(lib, lv, reps=20)
    run = bench.GraphedStep([lv])
    for _ in range(5):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    tot = sum(v[0] for v in st.values())
    print(f"level {i}: n {lv['n']:6d} rows {lv['n'] * lv['f']:7d} e {lv['e']:8d}  graph replay {ms:.3f} ms   sum of stages {tot:.3f} ms")
    print("   " + "  ".join(f"{k} {v[0] * 1e3:.0f}us" for k, v in sorted(st.items())))
