#!/usr/bin/env python3
"""The ResNetFormer block's glue alone (block minus its convolution) at the bench stack's level 0, a few steps: run under
`rocprofv3 --kernel-trace --stats` for the per-kernel view of scope row f-3.   usage: tools/profile_glue.py [level] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W

dev = torch.device("cuda", 0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lv = W.build_stack(W.WORKLOADS["headline"], dev, 0)[level]


class NoConv(torch.nn.Module):
    def forward(self, p_pc_in, p_pc_out, p_in_features, p_neighborhood):
        return p_in_features


c = lv["c"]
blk = amd.ResNetFormer(c, c, amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu"), amd.BatchNormPC, 0.1).to(dev)
blk.spatial_conv_ = NoConv()
blk.train()
x = torch.randn(lv["n"] * lv["f"], c, device=dev, requires_grad=True)
g = torch.randn(lv["n"] * lv["f"], c, device=dev)
for _ in range(steps):
    x.grad = None
    blk.zero_grad(set_to_none=True)
    blk(lv["pc"], x, lv["nbh"]).backward(g)
torch.cuda.synchronize()
print("done")
