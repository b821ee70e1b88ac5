#!/usr/bin/env python3
"""Host-side cost of one conv call: eager forward+backward wall time per level of the bench stack against the same
step replayed from a HIP graph (GPU time only).  The difference is Python + ctypes + allocator + autograd overhead."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from se3conv3d_amd import workloads as W

dev = torch.device("cuda", 0)
levels = W.build_stack(W.WORKLOADS["headline"], dev, 0)
for i, lv in enumerate(levels):
    for _ in range(5):
        bench.step([lv])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        bench.step([lv])
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50 * 1e3
    # host-only: time to ISSUE the step (no sync inside the loop), measured on a tiny level the GPU hides nothing of
    t0 = time.perf_counter()
    for _ in range(50):
        bench.step([lv])
    issue = (time.perf_counter() - t0) / 50 * 1e3
    torch.cuda.synchronize()
    g = bench.GraphedStep([lv])
    for _ in range(5):
        g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        g()
    torch.cuda.synchronize(); replay = (time.perf_counter() - t0) / 50 * 1e3
    print(f"level {i}: n {lv['n']:6d}  eager {eager:.3f} ms  (host issue time {issue:.3f} ms)  graph replay {replay:.3f} ms", flush=True)
