#!/usr/bin/env python3
"""Target program for rocprofv3 --kernel-trace --stats: PointcloudRotEquiv construction (16-NN + PCA frames + shuffle) on a
DFaust-sized level 0 (58 k points) and a small level (1.8 k points), 20 times each; prints the wall time per construction."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3conv3d_amd import pc as _pc, workloads as W

dev = torch.device("cuda:0")
pts, bid = W.faust_raw_batch(dev)
clouds = W.faust_clouds(pts, bid)
cfg = {"pca": True, "n_frames": 2, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
for lvl in (0, 3):
    p, b = clouds[lvl].pts_, clouds[lvl].batch_ids_
    for _ in range(3):
        _pc.PointcloudRotEquiv(p, b, cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        _pc.PointcloudRotEquiv(p, b, cfg)
    torch.cuda.synchronize()
    print(f"level {lvl}: {p.shape[0]} points, {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per construction")
