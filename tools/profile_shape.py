#!/usr/bin/env python3
"""Stage times (library HIP-event hooks) of one layer fwd+bwd at a given shape:  N F C_in C_out [degree]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import se3conv3d_amd as amd
from se3conv3d_amd.workloads import radius_for_degree
from se3conv3d_amd import _lib

n, f, ci, co = (int(v) for v in sys.argv[1:5])
deg = int(sys.argv[5]) if len(sys.argv) > 5 else 32
dev = torch.device("cuda", 0)
lib = _lib.load()
torch.manual_seed(0)
pts = torch.rand(n, 3, device=dev)
bid = torch.zeros(n, dtype=torch.int32, device=dev)
pc = amd.pc.PointcloudRotEquiv(pts, bid, {"pca": False, "n_frames": f, "fixed_axis": False})
r = radius_for_degree(n, deg)
nbh = amd.pc.BQNeighborhood(pc, pc, r)
conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(ci, co).to(dev)
conv.norm_neigh_dist_.fill_(1.0 / r)
conv.norm_num_neighs_.fill_(0.03)
lv = dict(pc=pc, nbh=nbh, conv=conv, x=torch.randn(n * f, ci, device=dev, requires_grad=True),
          g=torch.randn(n * f, co, device=dev), n=n, e=nbh.neighbors_.shape[0], r=r)
for _ in range(3):
    bench.step([lv])          # warm-up: lazy builds, allocator, clocks (the first call would dominate a 5-step average)
torch.cuda.synchronize()
st = bench.profile_level(lib, lv, 10)
print(f"N={n} F={f} C={ci}->{co} E={lv['e']}: sum {sum(v[0] for v in st.values()):.3f} ms")
print("   ", {k: round(v[0], 4) for k, v in sorted(st.items())})
