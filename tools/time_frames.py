#!/usr/bin/env python3
"""Timing of the neighbourhood / frame construction stages at the headline size (not part of the conv-only metric)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from oracle import se3conv_oracle as O
from se3conv3d_amd.workloads import radius_for_degree

def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for n in (65536, 150000):
    pts = torch.rand(n, 3, device="cuda"); bid = torch.zeros(n, dtype=torch.int32, device="cuda")
    r = radius_for_degree(n, 32)
    knn = amd.ops.knn_query(pts, bid, 16)
    print(f"N={n}: ball_query(k~32) {t(lambda: amd.ops.ball_query(pts, pts, bid, bid, r)):.3f} ms | "
          f"knn(16) grid {t(lambda: amd.ops.knn_query(pts, bid, 16, 1, 'grid')):.3f} ms, all-pairs {t(lambda: amd.ops.knn_query(pts, bid, 16, 1, 'scan'), 2):.3f} ms | pca_frames {t(lambda: amd.ops.pca_frames(pts, knn)):.3f} ms")
    import ctypes as C
    lib = amd._lib.load()
    lib.se3_profile_reset(); lib.se3_profile_enable(1)
    for _ in range(3): amd.ops.knn_query(pts, bid, 16, 1, 'grid')
    torch.cuda.synchronize(); lib.se3_profile_enable(0)
    st = {}
    for tag in (b"knn_sort", b"knn_cells", b"knn_fallback"):
        ms, cnt = C.c_double(0), C.c_int64(0)
        lib.se3_profile_read(tag, C.byref(ms), C.byref(cnt)); st[tag.decode()] = round(ms.value / max(cnt.value, 1), 3)
    mn_, mx_ = amd.ops.batch_aabb(pts, bid, 1)
    cell_ = amd.ops._knn_cell_size(mn_, mx_, torch.tensor([n], device="cuda"), 16)
    dk = (pts[knn[:, 15].long()] - pts).norm(dim=1)
    print("        grid kNN stages (ms):", st, "| cell", round(float(cell_), 4), "| queries left to the fallback:",
          int((dk >= 0.999 * cell_).sum()))
    nb, _ = amd.ops.ball_query(pts, pts, bid, bid, r)
    print(f"        csr_transpose {t(lambda: amd.ops.csr_transpose(nb, n)):.3f} ms, edges {nb.shape[0]}")
    pc = amd.pc.Pointcloud(pts, bid)
    cell = r / 2
    samp = amd.pc.GridSubSample(pc, cell)
    feats = torch.randn(n, 64, device="cuda")
    print(f"        GridSubSample(cell=r/2 -> {samp.num_out_} pts) build {t(lambda: amd.pc.GridSubSample(pc, cell)):.3f} ms | "
          f"avg pts {t(lambda: samp.__subsample_tensor__(pts, 'avg')):.3f} ms | avg feats[N,64] {t(lambda: samp.__subsample_tensor__(feats, 'avg')):.3f} ms | "
          f"max batch ids {t(lambda: samp.__subsample_tensor__(bid, 'max')):.3f} ms | upsample feats {t(lambda: samp.__upsample_tensor__(feats[:samp.num_out_])):.3f} ms")
    print(f"        PointHierarchy(3 sub-samples) {t(lambda: amd.pc.PointHierarchy(pc, 3, 'grid_avg', grid_radii=[cell, 2 * cell, 4 * cell])):.3f} ms")
