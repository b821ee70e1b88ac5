#!/usr/bin/env python3
"""GPU time of the self-k-NN (k = 16) of a DFaust-sized level 0 (58 k points, 32 bodies) and level 1 (34 k): HIP events around
20 calls of ops.knn_query with the boxes given.   usage: tools/time_knn.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W

dev = torch.device("cuda:0")
pts, bid = W.faust_raw_batch(dev)
clouds = W.faust_clouds(pts, bid)
for lvl in (0, 1, 2):
    pc = clouds[lvl]
    box = pc.aabb()
    for _ in range(3):
        amd.ops.knn_query(pc.pts_, pc.batch_ids_, 16, pc.num_batches(), box=box)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        amd.ops.knn_query(pc.pts_, pc.batch_ids_, 16, pc.num_batches(), box=box)
    e1.record()
    torch.cuda.synchronize()
    print(f"level {lvl}: {pc.pts_.shape[0]} points  {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per k-NN query (keys + sort + search)")
