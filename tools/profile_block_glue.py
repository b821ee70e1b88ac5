#!/usr/bin/env python3
"""Target program for rocprofv3 --kernel-trace --stats: the row-wise glue of the FAUST network's eight ResNetFormer blocks
(bench.py `faust_step`, part block_glue: the blocks with their convolution taken out), forward + backward, 20 times eagerly;
prints the wall time per repetition and the number of kernel launches per repetition."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W

dev = torch.device("cuda:0")
pts, bid = W.faust_raw_batch(dev)
clouds = W.faust_clouds(pts, bid)


class NoConv(torch.nn.Module):
    def forward(self, p_pc_in, p_pc_out, p_in_features, p_neighborhood):
        return p_in_features


fac = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
glue = []
for level, width in ((1, 32), (2, 64), (3, 128), (4, 256)):
    pc = clouds[level]
    rows = pc.pts_.shape[0] * 2
    for _ in range(2):
        blk = amd.ResNetFormer(width, width, fac, amd.BatchNormPC, 0.1).to(dev)
        blk.spatial_conv_ = NoConv()
        blk.train()
        glue.append((blk, pc, torch.randn(rows, width, device=dev, requires_grad=True), torch.randn(rows, width, device=dev)))


def all_glue():
    for blk, pc, x, g in glue:
        x.grad = None
        blk.zero_grad(set_to_none=True)
        blk(pc, x, None).backward(g)


for _ in range(3):
    all_glue()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    all_glue()
torch.cuda.synchronize()
print(f"block glue, eager: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per repetition (8 blocks, forward + backward)")

# the same as one captured graph (what bench.py's faust_step reports as block_glue)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        all_glue()
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    all_glue()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    graph.replay()
torch.cuda.synchronize()
print(f"block glue, one captured graph: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per replay")
