#!/usr/bin/env python3
"""ms per ResNetFormer block (forward + backward, training mode, drop path 0.1) with the library's fused row-wise
glue kernels against the plain torch modules, at the sizes of the bench stack's levels; also the glue alone (block
minus its convolution).  Eager launches and hipGraph replay."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from se3conv3d_amd import blocks, workloads as W

dev = torch.device("cuda", 0)
levels = W.build_stack(W.WORKLOADS["headline"], dev, 0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


class NoConv(torch.nn.Module):
    """stands in for the convolution when timing the glue alone"""
    def forward(self, p_pc_in, p_pc_out, p_in_features, p_neighborhood):
        return p_in_features


for lv in levels:
    c = lv["c"]
    fac = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
    blk = amd.ResNetFormer(c, c, fac, amd.BatchNormPC, 0.1).to(dev)
    blk.spatial_conv_.load_state_dict(lv["conv"].state_dict())
    blk.train()
    x = torch.randn(lv["n"] * lv["f"], c, device=dev, requires_grad=True)
    g = torch.randn(lv["n"] * lv["f"], c, device=dev)

    def step(b):
        x.grad = None
        b.zero_grad(set_to_none=True)
        b(lv["pc"], x, lv["nbh"]).backward(g)

    glue = amd.ResNetFormer(c, c, fac, amd.BatchNormPC, 0.1).to(dev)
    glue.spatial_conv_ = NoConv()
    glue.train()
    row = []
    for fused in (True, False):
        blocks.FUSED = fused
        for name, b in (("block", blk), ("glue only", glue)):
            eager = timed(lambda: step(b), 20)
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step(b)
            torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step(b)
            replay = timed(gr.replay, 50)
            row.append(f"{'fused' if fused else 'torch'} {name}: eager {eager:.3f} ms, graph {replay:.3f} ms")
    blocks.FUSED = True
    print(f"level n={lv['n']} rows={lv['n'] * lv['f']} C={c}: " + " | ".join(row), flush=True)
