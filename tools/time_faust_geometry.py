#!/usr/bin/env python3
"""Where the geometry of one DFaust step goes (bench.py `faust_step`, parts hierarchy_and_frames / neighbourhoods): wall time of
every stage of create_hierarchy (tasks/SemSeg/train_dfaust_rot.py:108-158) and of the network's neighbourhoods, eager."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import se3conv3d_amd as amd
from se3conv3d_amd import pc as _pc, workloads as W

dev = torch.device("cuda:0")
pts, bid = W.faust_raw_batch(dev)
cfg = {"pca": True, "n_frames": 2, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}


def timed(fn, reps=10):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


raw = _pc.Pointcloud(pts, bid)
ms, samp = timed(lambda: _pc.GridSubSample(raw, 0.04))
print(f"GridSubSample(raw {pts.shape[0]} pts, 0.04): {ms:.3f} ms")
ms, (p0, b0) = timed(lambda: (samp.__subsample_tensor__(raw.pts_, "avg"), samp.__subsample_tensor__(raw.batch_ids_, "max")))
print(f"  subsample pts + batch ids: {ms:.3f} ms -> {p0.shape[0]} points")
ms, pc0 = timed(lambda: _pc.PointcloudRotEquiv(p0, b0, cfg))
print(f"PointcloudRotEquiv(level 0: kNN 16 + PCA frames + shuffle): {ms:.3f} ms")
ms, hier = timed(lambda: _pc.PointHierarchyRotEquiv(pc0, 4, "grid_avg", grid_radii=[0.05, 0.1, 0.2, 0.4]))
print(f"PointHierarchyRotEquiv(4 levels, frames per level): {ms:.3f} ms -> {[c.pts_.shape[0] for c in hier.pcs_]}")
ms, samp_out = timed(lambda: _pc.GridSubSample(raw, 0.04, p_rnd_sample=True))
print(f"GridSubSample(raw, 0.04, rnd): {ms:.3f} ms")
ms, out_pc = timed(lambda: _pc.PointcloudRotEquiv(samp_out.__subsample_tensor__(raw.pts_, "avg"),
                                                   samp_out.__subsample_tensor__(raw.batch_ids_, "max"), cfg))
print(f"output cloud (subsample + frames): {ms:.3f} ms")
ms, _ = timed(lambda: W.faust_clouds(pts, bid))
print(f"faust_clouds total: {ms:.3f} ms")
clouds = list(hier.pcs_) + [out_pc]
calls = W.faust_network_calls(os.path.join(ROOT, "tests", "golden", "network_faust_calls.npz"))
nbhs = W.faust_neighbourhoods(clouds, calls)
caps = {k: int(nb.num_edges() * 1.25) + 64 for k, nb in nbhs.items()}


def fresh(call_list):
    # a step's clouds are new objects: no source grid survives from the step before (round 6: the queries of ONE step that
    # search a cloud with one radius share its grid)
    for c in clouds:
        amd.ops.forget_source_grids(c)
    return W.faust_neighbourhoods(clouds, call_list, caps)


for key in nbhs:
    one = [c for c in calls if (c["level_in"], c["level_out"], c["radius"]) == key][:1]
    ms, _ = timed(lambda: fresh(one))
    print(f"neighbourhood {key}: {ms:.3f} ms  ({nbhs[key].num_edges()} edges)")
ms, _ = timed(lambda: fresh(calls))
print(f"all {len(nbhs)} neighbourhoods: {ms:.3f} ms")
# the same with every query sorting its own source cloud (round 5's form), alternating
for rep in range(3):
    for on in (False, True):
        amd.ops.SHARED_GRIDS = on
        ms, _ = timed(lambda: fresh(calls))
        print(f"  all {len(nbhs)} neighbourhoods, source grids {'shared' if on else 'per query'}: {ms:.3f} ms")
