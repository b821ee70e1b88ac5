#!/usr/bin/env python3
"""Single-layer forward+backward time of the operator on shapes other than the headline one (hipGraph replay,
conv only): the configurations of the reference's task scripts -- ScanNet-like single frame, DFaust-like four
frames in batches, wider / narrower channels."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import se3conv3d_amd as amd
from se3conv3d_amd.workloads import radius_for_degree

dev = torch.device("cuda", 0)
CASES = [  # name, points per element, batch elements, frames, c_in, c_out, degree, fixed axis
    ("headline          ", 65536, 1, 2, 64, 64, 32, False),
    ("scannet-like F=1  ", 150000, 1, 1, 64, 64, 32, 2),
    ("dfaust-like  F=4  ", 4096, 16, 4, 32, 32, 24, False),
    ("wide  C=128       ", 65536, 1, 2, 128, 128, 32, False),
    ("first layer 3->32 ", 65536, 1, 2, 3, 32, 32, False),
    ("narrow C=32       ", 65536, 1, 2, 32, 32, 32, False),
]
for name, n_el, nb, f, ci, co, deg, axis in CASES:
    torch.manual_seed(0)
    n = n_el * nb
    pts = torch.rand(n, 3, device=dev)
    bid = torch.arange(nb, device=dev, dtype=torch.int32).repeat_interleave(n_el)
    cfg = {"pca": False, "n_frames": f, "fixed_axis": axis}
    pc = amd.pc.PointcloudRotEquiv(pts, bid, cfg)
    r = radius_for_degree(n_el, deg)
    nbh = amd.pc.BQNeighborhood(pc, pc, r)
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(ci, co).to(dev)
    conv.norm_neigh_dist_.fill_(1.0 / r)
    conv.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / max(nbh.neighbors_.shape[0], 1))
    x = torch.randn(n * f, ci, device=dev, requires_grad=True)
    g = torch.randn(n * f, co, device=dev)
    lv = dict(pc=pc, nbh=nbh, conv=conv, x=x, g=g, n=n, e=nbh.neighbors_.shape[0], r=r)
    run = bench.GraphedStep([lv])
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    ep = lv["e"] * f * f
    print(f"{name} N={n:7d} F={f} C={ci:3d}->{co:3d} E={lv['e']:8d} E'={ep:9d}: {ms:7.3f} ms  "
          f"{n / ms / 1e3:6.2f} Mpts/s  {ep / ms / 1e6:6.2f} G frame-edges/s")
