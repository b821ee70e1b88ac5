#!/usr/bin/env python3
"""Eager fwd+bwd of ONE level of a workload's stack, for `rocprofv3 --kernel-trace --stats` (per-kernel durations of the
small levels):   python3 tools/profile_small_levels.py <workload> <level> [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from se3conv3d_amd import workloads as W

wl, level = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
levels = W.build_stack(W.WORKLOADS[wl], torch.device("cuda:0"), 0)
lv = levels[level]
for _ in range(reps):
    bench.step([lv])
torch.cuda.synchronize()
print(f"{wl} level {level}: n {lv['n']} rows {lv['n'] * lv['f']} e {lv['e']} c {lv['c']}")
