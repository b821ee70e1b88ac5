#!/usr/bin/env python3
"""BASELINE configs 2 / 3 at the networks' own shapes: forward + backward of the 21 convolution calls the reference's
FPNSegUNetMLPGeluRotEqFAUST makes (call list recorded from the reference, tests/golden/network_faust_calls.npz) on a DFaust-sized
synthetic batch (32 bodies x 4096 sampled points -> 0.04 grid, hierarchy 0.05 .. 0.4, PCA frames, F = 2), or (--network scannet,
round 6) of the 32 calls of FPNSegUNetMLPGeluRotEqScanNet (tests/golden/network_scannet_calls.npz) on a ScanNet-sized batch
(6 synthetic rooms x 120 000 points -> 0.1 grid, hierarchy 0.2 .. 1.6, PCA frames about the up axis, F = 1).  Per call:
graph-replay time of forward + backward (dX except for the FAUST network's first convolution, dA, dbeta, dW); then all calls
as one captured graph.      usage: tools/time_network_convs.py [--network faust|scannet] [bodies | scenes]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import se3conv3d_amd as amd  # noqa: E402
from se3conv3d_amd import workloads as W  # noqa: E402

argv = sys.argv[1:]
network = "faust"
if argv[:1] == ["--network"]:
    network, argv = argv[1], argv[2:]
dev = torch.device("cuda", 0)
amd.set_precision(os.environ.get("SE3CONV_PRECISION", "bf16x3"))
if network == "faust":
    bodies = int(argv[0]) if argv else 32
    recs = W.build_faust_network_convs(dev, os.path.join(ROOT, "tests", "golden", "network_faust_calls.npz"), bodies=bodies)
    what = f"{bodies} bodies"
else:
    scenes = int(argv[0]) if argv else 6
    recs = W.build_scannet_network_convs(dev, os.path.join(ROOT, "tests", "golden", "network_scannet_calls.npz"), scenes=scenes)
    what = f"{scenes} rooms x 120000 points"
frames = recs[0]["f"]
levels = {}
for r in recs:
    levels[r["level_in"]] = r["n_in"]
    levels[r["level_out"]] = r["n_out"]
print(f"{network}: {what}; points per level (5 = output cloud): {dict(sorted(levels.items()))}")


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


from se3conv3d_amd import _lib  # noqa: E402

lib = _lib.load()
total = 0.0
for r in recs:
    run = bench.GraphedStep(None, fn=lambda _lv=None, r=r: bench.step_two_clouds(r))
    ms = timed(run)
    total += ms
    ab = W.layer_bytes_two_clouds(r["n_in"], r["n_out"], r["e"], frames, frames, r["c_in"], r["c_out"])
    print(f"{r['name']}: level {r['level_in']} -> {r['level_out']}  rows {r['n_in'] * frames:7d} -> {r['n_out'] * frames:7d}  edges {r['e']:8d}  "
          f"C {r['c_in']:3d} -> {r['c_out']:3d}  {ms:.3f} ms  layer_frac {ab / (ms * 1e-3) / 1e9 / 8000.0:.3f}")
    st = bench.profile_level(lib, None, 5, fn=lambda r=r: bench.step_two_clouds(r))
    print("        " + "  ".join(f"{k} {v[2] * 1e3:.0f}" for k, v in sorted(st.items())) + "  (us per step)")


clouds = {id(c): c for r in recs for c in (r["pc_in"], r["pc_out"])}


def all_calls(_lv=None):
    # one step of the network: the clouds' geometry records are built once (by the first convolution that touches each
    # cloud) and shared by the other calls, forward and backward
    for c in clouds.values():
        amd.ops.invalidate_prepared(c)
    for r in recs:
        bench.step_two_clouds(r)


for r in recs:
    r["own_clouds"] = False


ms_all = timed(bench.GraphedStep(None, fn=all_calls), reps=20)
n0 = levels[0]
print(f"sum of the {len(recs)} calls {total:.3f} ms; all {len(recs)} as one captured graph {ms_all:.3f} ms = {n0 / ms_all / 1e3:.2f} Mpoints/s of level-0 points "
      f"({n0} points, forward + backward of the network's convolutions alone)")
