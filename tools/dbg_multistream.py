#!/usr/bin/env python3
"""Diagnostic: capture of library calls on several streams of one graph (which combination breaks capture_end?).
usage: dbg_multistream.py <n_side_streams> <levels on side: e.g. 1,2,3> [points]"""
import faulthandler
import os
import sys

import torch

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd  # noqa: E402
from se3conv3d_amd import layers as L, ops as O, workloads as W  # noqa: E402

n_side = int(sys.argv[1])
side_levels = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else []
spec = dict(W.WORKLOADS["headline"])
if len(sys.argv) > 3:
    spec["points"] = int(sys.argv[3])
amd.set_precision("bf16x3")
dev = torch.device("cuda", 0)
levels = W.build_stack(spec, dev, seed=0)
print("rows per level", [lv["n"] * 2 for lv in levels], flush=True)
streams = [torch.cuda.Stream() for _ in range(n_side)]


def raw_step(lv):
    conv = lv["conv"]
    geom = L._geometry_of(lv["pc"], lv["pc"], lv["nbh"])
    with torch.no_grad():
        out, t_save = O.se3conv_forward(geom, lv["x"], conv.proj_axes_, conv.proj_biases_, conv.conv_weights_,
                                        conv.norm_neigh_dist_, conv.norm_num_neighs_, save_t=True)
        return O.se3conv_backward(geom, lv["x"], conv.proj_axes_, conv.proj_biases_, conv.conv_weights_,
                                  conv.norm_neigh_dist_, conv.norm_num_neighs_, t_save, lv["g"])


def step():
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)
    keep = []
    for i, lvl in enumerate(side_levels):
        with torch.cuda.stream(streams[i % max(n_side, 1)]) if n_side else torch.cuda.stream(cur):
            keep.append(raw_step(levels[lvl]))
    for lvl in range(4):
        if lvl not in side_levels:
            keep.append(raw_step(levels[lvl]))
    for st in streams:
        cur.wait_stream(st)
    return keep


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    step()
print("captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed ok", flush=True)
