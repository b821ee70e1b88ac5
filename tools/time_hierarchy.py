#!/usr/bin/env python3
"""Scope rows f-2 / f-3 on the GPU: wall time of the hierarchy build (3 grid sub-samples) and of the pool / up-sample /
frame-pool maps with their gradients, next to the torch-op formulation they replaced (unique + scatter_reduce + index)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import se3conv3d_amd as amd

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def torch_subsample(pts, bid, nb, cell):
    mn, mx = amd.ops.batch_aabb(pts, bid, nb)
    mn, mx = mn - 1e-6, mx + 1e-6
    nc = (((mx - mn) / cell).to(torch.int32) + 1).max(dim=0)[0]
    keys = amd.ops.compute_keys(pts, bid, mn, nc, torch.full((3,), cell, dtype=torch.float32, device=pts.device))
    _, ids = torch.unique(keys, return_inverse=True)
    m = int(ids.max().item()) + 1
    idx = ids[:, None].expand(-1, 3)
    new_pts = torch.zeros((m, 3), device=pts.device).scatter_reduce(0, idx, pts, "mean", include_self=False)
    new_bid = torch.zeros(m, dtype=bid.dtype, device=pts.device).scatter_reduce(0, ids, bid, "amax", include_self=False)
    return ids, m, new_pts, new_bid


for n in (65536, 150000):
    torch.manual_seed(0)
    pts = torch.rand(n, 3, device=dev)
    bid = torch.zeros(n, dtype=torch.int32, device=dev)
    r0 = (3.0 * 32 / (4.0 * 3.141592653589793 * n)) ** (1.0 / 3.0)
    radii = [r0, 2 * r0, 4 * r0]

    def build():
        p, b = pts, bid
        for r in radii:
            c = amd.ops.grid_subsample(p, b, r, 1)
            p, b = c.pts, c.batch_ids

    def build_torch():
        p, b = pts, bid
        for r in radii:
            _, _, p, b = torch_subsample(p, b, 1, r)

    print(f"N={n}: hierarchy build (3 sub-samples)  library {timed(build):.3f} ms   torch unique+scatter {timed(build_torch):.3f} ms")
    cells = amd.ops.grid_subsample(pts, bid, r0, 1)
    ids64 = cells.cell_ids.to(torch.int64)
    m, c = cells.n_cells, 64
    x = torch.randn(n, c, device=dev, requires_grad=True)
    g = torch.randn(m, c, device=dev)
    for method, how in (("avg", "mean"), ("max", "amax")):
        def lib_pool():
            x.grad = None
            amd.ops.GridPool.apply(x, cells, method).backward(g)

        def torch_pool():
            x.grad = None
            torch.zeros((m, c), device=dev).scatter_reduce(0, ids64[:, None].expand(-1, c), x, how, include_self=False).backward(g)

        t_lib, t_torch = timed(lib_pool), timed(torch_pool)
        gb = (2 * n * c * 4 + 2 * m * c * 4) / 1e9
        print(f"  pool_tensor {method} [N,{c}] fwd+bwd ({m} cells): library {t_lib:.3f} ms ({gb / t_lib * 1e3:.0f} GB/s)   torch scatter_reduce {t_torch:.3f} ms")
    z = torch.randn(m, c, device=dev, requires_grad=True)
    gu = torch.randn(n, c, device=dev)

    def lib_up():
        z.grad = None
        amd.ops.GridUpsample.apply(z, cells).backward(gu)

    def torch_up():
        z.grad = None
        z[ids64].backward(gu)

    print(f"  upsample_tensor fwd+bwd: library {timed(lib_up):.3f} ms   torch index {timed(torch_up):.3f} ms")
    f = 2
    xf = torch.randn(n * f, c, device=dev, requires_grad=True)
    gf = torch.randn(n, c, device=dev)
    for method in ("avg", "max"):
        def lib_fp():
            xf.grad = None
            amd.ops.FramePool.apply(xf, f, method).backward(gf)

        def torch_fp():
            xf.grad = None
            v = xf.reshape(n, f, c)
            (v.mean(1) if method == "avg" else v.max(1)[0]).backward(gf)

        t_lib, t_torch = timed(lib_fp), timed(torch_fp)
        gb = (2 * n * f * c * 4 + 2 * n * c * 4) / 1e9
        print(f"  feature_pooling {method} F={f} fwd+bwd: library {t_lib:.3f} ms ({gb / t_lib * 1e3:.0f} GB/s)   torch {t_torch:.3f} ms")
