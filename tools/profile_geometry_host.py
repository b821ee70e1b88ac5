#!/usr/bin/env python3
"""Host side of the eager geometry of a DFaust step under cProfile: 20 x the 15 neighbourhoods (source grids forgotten per
repetition, as a new step's clouds have none), then 20 x create_hierarchy -- where the Python above the C ABI spends its time
(the builds are bound by the host's launch rate).   usage: tools/profile_geometry_host.py"""
import os, sys, time, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import se3conv3d_amd as amd
from se3conv3d_amd import pc as _pc, workloads as W
dev = torch.device("cuda:0")
pts, bid = W.faust_raw_batch(dev)
clouds = W.faust_clouds(pts, bid)
calls = W.faust_network_calls(os.path.join(ROOT, "tests", "golden", "network_faust_calls.npz"))
nbhs = W.faust_neighbourhoods(clouds, calls)
caps = {k: int(nb.num_edges() * 1.25) + 64 for k, nb in nbhs.items()}
def fresh():
    for c in clouds:
        amd.ops.forget_source_grids(c)
    return W.faust_neighbourhoods(clouds, calls, caps)
for _ in range(5): fresh()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): fresh()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
pr = cProfile.Profile()
pr.enable()
for _ in range(20): W.faust_clouds(pts, bid)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
