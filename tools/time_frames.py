#!/usr/bin/env python3
"""Timing of the neighbourhood / frame construction stages at the headline size (not part of the conv-only metric)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from oracle import se3conv_oracle as O

def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for n in (65536, 150000):
    pts = torch.rand(n, 3, device="cuda"); bid = torch.zeros(n, dtype=torch.int32, device="cuda")
    r = O.radius_for_degree(n, 32)
    knn = amd.ops.knn_query(pts, bid, 16)
    print(f"N={n}: ball_query(k~32) {t(lambda: amd.ops.ball_query(pts, pts, bid, bid, r)):.3f} ms | "
          f"knn(16) {t(lambda: amd.ops.knn_query(pts, bid, 16)):.3f} ms | pca_frames {t(lambda: amd.ops.pca_frames(pts, knn)):.3f} ms")
    nb, _ = amd.ops.ball_query(pts, pts, bid, bid, r)
    print(f"        csr_transpose {t(lambda: amd.ops.csr_transpose(nb, n)):.3f} ms, edges {nb.shape[0]}")
