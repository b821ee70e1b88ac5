#!/usr/bin/env python3
"""Per-level stage times (library HIP-event hooks) + eager/graph wall time of every level of the bench stack."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import se3conv3d_amd as amd
from oracle import se3conv_oracle as O
from se3conv3d_amd import _lib

lib = _lib.load()
amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
levels = bench.build_stack(amd, O, torch.device("cuda", 0), 0)
for i, lv in enumerate(levels):
    st = bench.profile_level0(lib, lv, 5)
    g = bench.GraphedStep([lv])
    for _ in range(3): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    print(f"level {i}: n={lv['n']} e={lv['e']} graph_ms={dt:.3f} sum_stages={sum(v[0] for v in st.values()):.3f}")
    print("   ", {k: round(v[0], 4) for k, v in sorted(st.items())})
