#!/usr/bin/env python3
"""Capacity-bounded neighbourhood builds of the bench stack's levels, repeated: the target program for
`rocprofv3 --kernel-trace --stats` (which kernels the 0.39 ms of the end-to-end leg are made of)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W
dev = torch.device("cuda:0")
levels = W.build_stack(W.WORKLOADS["headline"], dev, 0)
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for i, lv in enumerate(levels):
    if only >= 0 and i != only:
        continue
    cap = int(lv["e"] * 1.25) + 64
    def build():
        nb = amd.pc.BQNeighborhood(lv["pc"], lv["pc"], lv["r"], p_capacity=cap)
        amd.layers._geometry_of(lv["pc"], lv["pc"], nb).transpose()
    for _ in range(3): build()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): build()
    torch.cuda.synchronize()
    print(f"level {i}: n {lv['n']} bounded build {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
