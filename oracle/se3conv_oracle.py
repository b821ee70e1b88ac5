"""CPU oracle for the PNEConvLayerRotEquiv hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU / numpy restatement of the reference's algorithm for
the one path this repository accelerates.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
Nothing under ``se3conv3d_amd/`` imports it, and the product path raises if the HIP
library is missing instead of falling back to this file.

Pinning (see DESIGN.md "Oracle"):
  * descriptor construction, frame-pair ordering, kernel MLP + exact-erf GELU, einsum
    contraction, the two scalings, parameter init and the EMA normalisers are pinned by
    golden fixtures generated from the reference's own Python (``tools/gen_golden.py``
    imports ``/root/reference/point_cloud_lib`` with stand-ins for the three native
    modules that are not installed; fixtures under ``tests/golden/``).
  * the arithmetic of the two CUDA ops (``feat_basis_proj{,_grad}``) and of the CUDA
    ball query cannot be executed here (no nvcc, no GPU): for those this file follows the
    CUDA sources line by line (citations below) and their semantics are pinned only by the
    reference's Python call sites -> "parity unpinned" for the native arithmetic itself.

Conventions:
  rows of feature tensors are ``point * F + frame``; frames are ``[N, F, 9]`` row-major
  3x3 with COLUMNS = basis vectors; neighbours are ``[E, 2]`` (col0 = sample/out point,
  col1 = source/in point) sorted by col0; ``ends`` are inclusive end offsets per sample.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------------------------
# a10: ball query  (PCL/custom_ops/BallQuery.py:34-40, OPS/ball_query/*.cu)
# ----------------------------------------------------------------------------------------------


def ball_query_grid_params(pts_src: torch.Tensor, batch_src: torch.Tensor, radius: float):
    """Per-batch AABB minimum and the grid size, as BallQuery.forward builds them.

    Follows PCL/custom_ops/BallQuery.py:34-38: min/max per batch id each shifted by -1e-6,
    ``num_cells = int((max - min) / radius) + 1`` with the maximum taken over batches.
    """
    b = int(batch_src.max().item()) + 1
    idx = batch_src.to(torch.int64)
    d = pts_src.shape[1]
    mn = torch.full((b, d), float("inf"), dtype=pts_src.dtype)
    mx = torch.full((b, d), float("-inf"), dtype=pts_src.dtype)
    mn = mn.scatter_reduce(0, idx[:, None].expand(-1, d), pts_src, "amin")
    mx = mx.scatter_reduce(0, idx[:, None].expand(-1, d), pts_src, "amax")
    mn = mn - 1e-6
    mx = mx - 1e-6
    num_cells = ((mx - mn) / radius).to(torch.int32) + 1
    num_cells = num_cells.max(dim=0)[0]
    return mn.to(torch.float32), num_cells.to(torch.int32)


def compute_keys(pts, batch_ids, aabb_min, num_cells, cell_size) -> torch.Tensor:
    """Grid cell key per point: OPS/ball_query/compute_keys.cu:32-71.

    cell = clamp(floor((p - aabbMin[b]) * (1/cell)), 0, n-1)   (grid_utils.cuh:57-67)
    key  = ((b*n0 + c0)*n1 + c1)*n2 + c2                          (grid_utils.cuh:79-93)
    ``cell_size`` is a length-D tensor; the reciprocal is taken in fp32 like
    compute_keys.cu:112 (torch::reciprocal) and multiplied, not divided.
    """
    pts = pts.to(torch.float32)
    inv = torch.reciprocal(cell_size.to(torch.float32))
    rel = (pts - aabb_min.to(torch.float32)[batch_ids.to(torch.int64)]) * inv
    cell = torch.floor(rel).to(torch.int64)
    nc = num_cells.to(torch.int64)
    cell = torch.minimum(torch.maximum(cell, torch.zeros_like(cell)), nc[None, :] - 1)
    key = torch.zeros(pts.shape[0], dtype=torch.int64)
    accum = 1
    for i in range(pts.shape[1] - 1, -1, -1):
        key = key + cell[:, i] * accum
        accum = accum * int(nc[i])
    return key + accum * batch_ids.to(torch.int64)


def ball_query(pts_src, pts_dst, batch_src, batch_dst, radius: float, chunk: int = 2048):
    """Radius neighbours, brute force, with the CUDA predicate.

    Predicate (OPS/ball_query/count_neighbors.cu:84-89, math_helper.cuh:316-320):
    ``sqrt(sum(((s - p) * (1/r))**2)) < 1`` in fp32, same batch id; the candidate set of
    the grid search (3x3 pencils x [key-1, key+1] in z, find_ranges_grid_ds.cu:41-166) is
    a superset of that ball for cell size = radius, so brute force gives the same edge SET.
    Output follows ball_query.cu:92-101 / store_neighbors.cu: ``neighbors[E,2]`` int64 with
    col0 = sample id, col1 = source id, grouped by sample; ``ends[M]`` int32 inclusive end
    offsets.  Order inside a sample is undefined in the reference (atomics); here it is
    ascending source id (canonical form used by the tests).
    """
    pts_src = pts_src.to(torch.float32)
    pts_dst = pts_dst.to(torch.float32)
    inv_r = torch.reciprocal(torch.tensor(radius, dtype=torch.float32))
    m = pts_dst.shape[0]
    rows, cols = [], []
    counts = torch.zeros(m, dtype=torch.int64)
    for s0 in range(0, m, chunk):
        s1 = min(m, s0 + chunk)
        d = (pts_dst[s0:s1, None, :] - pts_src[None, :, :]) * inv_r
        dist = torch.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
        hit = (dist < 1.0) & (batch_dst[s0:s1, None] == batch_src[None, :])
        r, c = torch.nonzero(hit, as_tuple=True)  # row-major: sorted by sample then source
        rows.append(r + s0)
        cols.append(c)
        counts[s0:s1] = hit.sum(1)
    neighbors = torch.stack((torch.cat(rows), torch.cat(cols)), dim=1).to(torch.int64)
    ends = torch.cumsum(counts, 0).to(torch.int32)
    return neighbors, ends


def ball_query_grid(pts_src, pts_dst, batch_src, batch_dst, radius: float):
    """The reference's grid algorithm restated with python loops (small inputs only).

    keys -> argsort -> pencil table -> 9 (dx,dy) offsets x z-window [cz-1, cz+1] clamped to
    the grid -> distance test.  (ball_query.cu:22-103, build_grid_ds.cu:31-66,
    find_ranges_grid_ds.cu:41-166, count_neighbors.cu:36-103.)  Used by the tests to show
    that the candidate windows lose no neighbour w.r.t. ``ball_query`` above.
    """
    mn, nc = ball_query_grid_params(pts_src, batch_src, radius)
    cs = torch.full((3,), radius, dtype=torch.float32)
    keys = compute_keys(pts_src, batch_src, mn, nc, cs)
    order = torch.argsort(keys, stable=True)
    skeys = keys[order].numpy()
    spts = pts_src[order].to(torch.float32).numpy()
    n0, n1, n2 = (int(v) for v in nc)
    inv_r = np.float32(1.0) / np.float32(radius)
    dkeys = compute_keys(pts_dst, batch_dst, mn, nc, cs).numpy()
    out = []
    counts = np.zeros(pts_dst.shape[0], dtype=np.int64)
    dpts = pts_dst.to(torch.float32).numpy()
    for s in range(pts_dst.shape[0]):
        key = int(dkeys[s])
        cz = key % n2
        cy = (key // n2) % n1
        cx = (key // (n2 * n1)) % n0
        b = key // (n2 * n1 * n0)
        found = []
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                x, y = cx + dx, cy + dy
                if x < 0 or x >= n0 or y < 0 or y >= n1:
                    continue
                z0, z1 = max(cz - 1, 0), min(cz + 1, n2 - 1)
                base = ((b * n0 + x) * n1 + y) * n2
                lo = np.searchsorted(skeys, base + z0, side="left")
                hi = np.searchsorted(skeys, base + z1, side="right")
                for j in range(lo, hi):
                    d = (dpts[s] - spts[j]) * inv_r
                    if np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) < np.float32(1.0):
                        found.append(int(order[j]))
        found.sort()
        counts[s] = len(found)
        out.extend((s, p) for p in found)
    neighbors = torch.tensor(out, dtype=torch.int64).reshape(-1, 2)
    return neighbors, torch.from_numpy(np.cumsum(counts)).to(torch.int32)


# ----------------------------------------------------------------------------------------------
# a11: frames  (PCL/pc/RotationFunctions.py:57-88 quaternion_to_matrix, :176-216 random rotations)
# ----------------------------------------------------------------------------------------------


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """Real-part-first quaternions -> 3x3 rotation matrices (RotationFunctions.py:57-88)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def random_frames(n: int, f: int, generator: torch.Generator | None = None) -> torch.Tensor:
    """``sample_reference_frames`` without a fixed axis (RotationFunctions.py:428-452):
    normalised N(0,1) quaternions with non-negative real part -> ``[n, f, 9]``."""
    o = torch.randn((n * f, 4), generator=generator, dtype=torch.float32)
    s = (o * o).sum(1)
    sign = torch.where(o[:, 0] < 0, -1.0, 1.0)
    o = o / (torch.sqrt(s) * sign)[:, None]
    return quaternion_to_matrix(o).reshape(n, f, 9)


def knn_query(pts: torch.Tensor, batch_ids: torch.Tensor, k: int) -> torch.Tensor:
    """Exact self-kNN inside each batch element (OPS/knn_query/knn_query.cu:18-132): the point itself first,
    ascending squared distance, ties to the lower index (strict ``>`` at :68), ``-1`` padded.  ``[N,k]`` int32."""
    pts = pts.to(torch.float32)
    n = pts.shape[0]
    out = torch.full((n, k), -1, dtype=torch.int32)
    for b in torch.unique(batch_ids):
        idx = torch.nonzero(batch_ids == b)[:, 0]
        p = pts[idx]
        d = p[:, None, :] - p[None, :, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
        order = torch.argsort(d2, dim=1, stable=True)[:, :k]
        out[idx, : order.shape[1]] = idx[order].to(torch.int32)
    return out


def sample_reference_frames_pca(points: torch.Tensor, knn_ids: torch.Tensor, axis_fixed=False) -> torch.Tensor:
    """PCA frames (PCL/pc/RotationFunctions.py:307-406) from fixed-k neighbour ids ``[N,k]`` (-1 = missing ->
    the point itself, :314-317).  Free case: covariance of the centred neighbours, ``torch.linalg.eigh``
    (ascending), whole matrix negated where det < 0, then the four column sign patterns with product +1
    -> ``[N,4,9]``.  Fixed axis (1 or 2): that coordinate zeroed, eigenpairs flipped to descending, patterns
    (1,1,1), (-1,-1,1), column permutation for axis 1, |x| < 1e-6 -> 0 -> ``[N,2,9]``."""
    n, k = knn_ids.shape
    ids = knn_ids.to(torch.int64).clone()
    rows = torch.arange(n)[:, None].expand(-1, k)
    ids[ids < 0] = rows[ids < 0]
    nm = points[ids].clone()  # [N,k,3]
    fixed = bool(axis_fixed)
    if fixed:
        nm[:, :, int(axis_fixed)] = 0
    nm = nm - nm.mean(dim=1, keepdim=True)
    c = torch.einsum("bij,bjk->bik", nm.transpose(1, 2), nm)
    _, vec = torch.linalg.eigh(c)
    if fixed:
        vec = torch.flip(vec, dims=[-1])
    vec = vec.clone()
    vec[torch.linalg.det(vec) < 0] *= -1
    pats = [(1, 1, 1), (-1, -1, 1)] if fixed else [(1, 1, 1), (1, -1, -1), (-1, 1, -1), (-1, -1, 1)]
    pm = torch.tensor(pats, dtype=vec.dtype)[:, None, :]  # [P,1,3]: scales the columns
    frames = pm[None] * vec[:, None]
    if fixed:
        if int(axis_fixed) == 1:
            frames = frames[:, :, :, [0, 2, 1]]
        frames = torch.where(frames.abs() < 1e-6, torch.zeros_like(frames), frames)
    return frames.reshape(n, len(pats), 9)


# ----------------------------------------------------------------------------------------------
# a1: frame-edge list and 9-D descriptors  (PCL/layers/PNEConvLayerRotEquiv.py:62-128)
# ----------------------------------------------------------------------------------------------


def edge_descriptors(pts_in, pts_out, frames_in, frames_out, neighbors, rho) -> torch.Tensor:
    """``[E, F_out*F_in, 9]`` descriptors in the reference's frame-pair order a*F_in + b.

    loc  = (rho * (x_p - y_s)) @ R_{s,a}                (:68-69, RotationFunctions.py:637-665)
    rel6 = rows 0,1 of R_{s,a}^T @ R_{p,b}             (:82-84, RotationFunctions.py:549-600,
                                                         matrix_to_rotation_6d :236-252)
    """
    s = neighbors[:, 0].to(torch.int64)
    p = neighbors[:, 1].to(torch.int64)
    f_out, f_in = frames_out.shape[1], frames_in.shape[1]
    rel = (pts_in[p] - pts_out[s]) * rho  # [E,3]
    r_out = frames_out[s].reshape(-1, f_out, 3, 3)
    r_in = frames_in[p].reshape(-1, f_in, 3, 3)
    loc = torch.matmul(rel[:, None, None, :], r_out).squeeze(2)  # [E,F_out,3]
    loc = loc[:, :, None, :].expand(-1, -1, f_in, -1)
    relrot = torch.matmul(r_out.transpose(2, 3)[:, :, None], r_in[:, None])  # [E,Fo,Fi,3,3]
    rel6 = relrot[..., :2, :].reshape(-1, f_out, f_in, 6)
    return torch.cat((loc, rel6), -1).reshape(-1, f_out * f_in, 9)


def get_rot_tensors(pts_in, pts_out, frames_in, frames_out, neighbors, rho, n_rows=None,
                    rel_rot: str = "6D") -> Dict[str, torch.Tensor]:
    """Restatement of ``PNEConvLayerRotEquiv.get_rot_tenors`` (:62-128).

    Returns ``rel_pts_rel_orient [E',9]``, ``neighbs [E',2]`` (col0 = s*F_out+a,
    col1 = p*F_in+b) sorted by col0 (stable here; the reference's sort is unstable so only
    the multiset per output row is defined) and ``neighbs_start_ids`` = inclusive cumsum of
    the per-row degree.  ``n_rows=None`` reproduces the reference quirk (:111-114: no
    ``dim_size`` => trailing rows without edges are dropped); pass ``N_out*F_out`` for the
    fixed length the HIP path uses.
    """
    f_out, f_in = frames_out.shape[1], frames_in.shape[1]
    desc = edge_descriptors_rel(pts_in, pts_out, frames_in, frames_out, neighbors, rho, rel_rot)
    desc = desc.reshape(-1, desc.shape[-1])  # D = 9 ("6D"), 12 ("matrix") or 7 ("quaternion")
    e = neighbors.shape[0]
    a = torch.arange(f_out).repeat_interleave(f_in).repeat(e)
    b = torch.arange(f_in).repeat(f_out).repeat(e)
    nb = neighbors.to(torch.int64).repeat_interleave(f_out * f_in, dim=0)
    nb = torch.stack((nb[:, 0] * f_out + a, nb[:, 1] * f_in + b), 1)
    order = torch.sort(nb[:, 0], stable=True).indices
    nb = nb[order]
    desc = desc[order]
    rows = int(nb[:, 0].max().item()) + 1 if n_rows is None else n_rows
    deg = torch.bincount(nb[:, 0], minlength=rows)
    ends = torch.cumsum(deg, 0).to(torch.int32)
    return {"rel_pts_rel_orient": desc, "neighbs": nb, "neighbs_start_ids": ends}


# ----------------------------------------------------------------------------------------------
# a2: kernel MLP  (PNEConvLayerRotEquiv.py:199-203; activation table PNEConvLayer.py:91-100)
# ----------------------------------------------------------------------------------------------


def kernel_mlp(desc, proj_axes, proj_biases, act: str = "gelu") -> torch.Tensor:
    """``act(desc @ A + beta)``; the activation table of PNEConvLayer.py:91-100: "gelu" = torch.nn.GELU() default
    (exact erf form, every *_rot configuration), "relu", "sin", "softmax" (over the basis functions), "linear"."""
    pre = torch.matmul(desc, proj_axes) + proj_biases
    if act == "gelu":
        return torch.nn.functional.gelu(pre)
    if act == "relu":
        return torch.relu(pre)
    if act == "sin":
        return torch.sin(pre)
    if act == "softmax":
        return torch.softmax(pre, dim=-1)
    if act == "linear":
        return pre
    raise ValueError(act)


# ----------------------------------------------------------------------------------------------
# a4/a5: feat_basis_proj and its gradient  (OPS/feature_aggregation/feat_basis_proj{,_grads}.cu)
# ----------------------------------------------------------------------------------------------


def feat_basis_proj(pt_basis, pt_features, neighbors, ends) -> torch.Tensor:
    """``T[m,c,k] = sum_{e in [ends[m-1], ends[m])} basis[e,k] * feat[neighbors[e,1], c]``.

    feat_basis_proj.cu:55-117 (segment bounds :59-61, product :97-99, layout
    ``m*C*K + c*K + k`` :115-116).  Rows are defined by ``ends`` alone; ``neighbors[:,0]``
    is not read by the kernel.
    """
    m = ends.shape[0]
    e = pt_basis.shape[0]
    seg = torch.repeat_interleave(
        torch.arange(m), torch.diff(ends.to(torch.int64), prepend=torch.zeros(1, dtype=torch.int64))
    )
    assert seg.shape[0] == e, "ends must cover every edge"
    contrib = pt_features[neighbors[:, 1].to(torch.int64)][:, :, None] * pt_basis[:, None, :]
    out = torch.zeros((m, pt_features.shape[1], pt_basis.shape[1]), dtype=pt_features.dtype)
    out.index_add_(0, seg, contrib)
    return out


def feat_basis_proj_grad(pt_basis, pt_features, neighbors, ends, grad_t):
    """Gradients of ``feat_basis_proj`` (feat_basis_proj_grads.cu:91-143).

    gBasis[e,k] = sum_c gT[m,c,k] * feat[p,c]        (:113-119, atomics :126)
    gFeat[p,c] += sum_k gT[m,c,k] * basis[e,k]        (:129-140)
    Returns ``(gFeat, gBasis)`` in the order of the native op (ops_list / FeatBasisProj.py:56).
    """
    m = ends.shape[0]
    seg = torch.repeat_interleave(
        torch.arange(m), torch.diff(ends.to(torch.int64), prepend=torch.zeros(1, dtype=torch.int64))
    )
    src = neighbors[:, 1].to(torch.int64)
    g = grad_t[seg]  # [E,C,K]
    g_basis = torch.einsum("eck,ec->ek", g, pt_features[src])
    g_feat = torch.zeros_like(pt_features)
    g_feat.index_add_(0, src, torch.einsum("eck,ek->ec", g, pt_basis))
    return g_feat, g_basis


# ----------------------------------------------------------------------------------------------
# a6/a7: the layer  (PNEConvLayerRotEquiv.py:160-216)
# ----------------------------------------------------------------------------------------------


def conv_forward(pts_in, pts_out, frames_in, frames_out, neighbors, feat, proj_axes, proj_biases,
                 conv_weights, rho, nu, pad_rows: bool = True, act: str = "gelu", rel_rot: str = "6D") -> torch.Tensor:
    """One ``PNEConvLayerRotEquiv`` forward ("mlp_*" branch, :178-216), differentiable
    w.r.t. ``feat, proj_axes, proj_biases, conv_weights`` through torch autograd.

    ``pad_rows=True`` returns ``N_out*F_out`` rows (HIP path / fixed quirk 1);
    ``False`` reproduces the reference's dropped trailing rows.
    """
    f_out, f_in = frames_out.shape[1], frames_in.shape[1]
    with torch.no_grad():
        rt = get_rot_tensors(pts_in, pts_out, frames_in, frames_out, neighbors, rho,
                             n_rows=pts_out.shape[0] * f_out if pad_rows else None, rel_rot=rel_rot)
    phi = kernel_mlp(rt["rel_pts_rel_orient"], proj_axes, proj_biases, act)
    seg = rt["neighbs"][:, 0]
    src = rt["neighbs"][:, 1]
    rows = rt["neighbs_start_ids"].shape[0]
    contrib = feat[src][:, :, None] * phi[:, None, :]
    t = torch.zeros((rows, feat.shape[1], phi.shape[1]), dtype=feat.dtype).index_add(0, seg, contrib)
    out = torch.einsum("nik,iko->no", t, conv_weights)
    out = out / f_in
    return out * nu


def conv_forward_backward(pts_in, pts_out, frames_in, frames_out, neighbors, feat, proj_axes,
                          proj_biases, conv_weights, rho, nu, grad_out, dtype=torch.float32, act: str = "gelu",
                          rel_rot: str = "6D"):
    """Forward + autograd backward in ``dtype``; returns ``out, dX, dA, dbeta, dW``."""
    cast = lambda t: t.detach().to(dtype)
    x = cast(feat).requires_grad_(True)
    a = cast(proj_axes).requires_grad_(True)
    b = cast(proj_biases).requires_grad_(True)
    w = cast(conv_weights).requires_grad_(True)
    out = conv_forward(cast(pts_in), cast(pts_out), cast(frames_in), cast(frames_out), neighbors, x, a, b, w,
                       cast(torch.as_tensor(rho)), cast(torch.as_tensor(nu)), act=act, rel_rot=rel_rot)
    out.backward(cast(grad_out))
    return out.detach(), x.grad, a.grad, b.grad, w.grad


def conv_forward_edgewise(pts_in, pts_out, frames_in, frames_out, neighbors, feat, proj_axes,
                          proj_biases, conv_weights, rho, nu) -> torch.Tensor:
    """Memory-lean forward (no ``[E',C,K]`` temporary) used for big CPU baselines: loops over
    frame pairs and uses a per-row ``[rows, C*K]`` accumulator; same arithmetic as
    ``conv_forward`` up to summation order."""
    f_out, f_in = frames_out.shape[1], frames_in.shape[1]
    n_out = pts_out.shape[0]
    c, k = feat.shape[1], proj_axes.shape[1]
    desc = edge_descriptors(pts_in, pts_out, frames_in, frames_out, neighbors, rho)
    s = neighbors[:, 0].to(torch.int64)
    p = neighbors[:, 1].to(torch.int64)
    t = torch.zeros((n_out * f_out, c * k), dtype=feat.dtype)
    for a in range(f_out):
        for b in range(f_in):
            phi = kernel_mlp(desc[:, a * f_in + b], proj_axes, proj_biases)
            contrib = (feat[p * f_in + b][:, :, None] * phi[:, None, :]).reshape(-1, c * k)
            t.index_add_(0, s * f_out + a, contrib)
    out = t @ conv_weights.reshape(c * k, -1)
    return out / f_in * nu


# ----------------------------------------------------------------------------------------------
# a8/a9: EMA normalisers and parameter init
# ----------------------------------------------------------------------------------------------


def ema_update(norm_neigh_dist, norm_num_neighs, radius: float, n_samples: int, n_edges: int):
    """One pre-process step for a ball-query neighbourhood (IConvLayer.py:76-97):
    rho <- 0.9 rho + 0.1 / r ; nu <- 0.9 nu + 0.1 * M / E  (point-level M and E)."""
    new_rho = torch.tensor(1.0 / radius, dtype=torch.float32)
    rho = 0.9 * norm_neigh_dist + 0.1 * new_rho
    new_nu = torch.tensor(n_samples / n_edges, dtype=torch.float32)
    nu = 0.9 * norm_num_neighs + 0.1 * new_nu
    return rho, nu


def init_parameters(dims: int, c_in: int, c_out: int, num_basis: int, generator=None):
    """Parameter init (PNEConvLayer.py:79-88, 151-158): A ~ U(+-sqrt(1/D)), beta = 0,
    W ~ U(+-sqrt(1/(C_in*K)))."""
    sa = math.sqrt(1.0 / dims)
    a = (torch.rand((dims, num_basis), generator=generator) * 2 - 1) * sa
    b = torch.zeros(num_basis)
    sw = math.sqrt(1.0 / (c_in * num_basis))
    w = (torch.rand((c_in, num_basis, c_out), generator=generator) * 2 - 1) * sw
    return a, b, w


# ----------------------------------------------------------------------------------------------
# synthetic workload shared by tests and bench
# ----------------------------------------------------------------------------------------------


def radius_for_degree(n: int, k: float) -> float:
    """r = (3k / (4 pi N))^(1/3): expected interior degree k for N uniform points in [0,1)^3."""
    return (3.0 * k / (4.0 * math.pi * n)) ** (1.0 / 3.0)


def pne_conv_forward_backward(pts_in, pts_out, neighbors, ends, x, proj_axes, proj_biases, conv_weights, rho, nu,
                              grad_out, dtype=torch.float64):
    """The reference's NON-equivariant ``PNEConvLayer`` ('mlp_gelu', 'add'; scope row f-4), restated from
    ``layers/PNEConvLayer.py:161-229`` and ``custom_ops/PNE.py:36-40``:

        basis[e]  = GELU(rho * (x_src(e) - y_smp(e)) @ A + beta)                       (LinearPNE + torch.nn.GELU())
        T[m,i,k]  = sum_{e in seg m} f[src(e), i] * basis[e, k]                         (FeatBasisProj)
        out       = einsum('nik,iko->no', T, W) * nu

    Returns out and the gradients (dx, dA, dbeta, dW) for ``grad_out`` by autograd on the restated forward."""
    t = lambda v: torch.as_tensor(v).to(dtype)  # noqa: E731
    x = t(x).clone().requires_grad_(True)
    a = t(proj_axes).clone().requires_grad_(True)
    b = t(proj_biases).clone().requires_grad_(True)
    w = t(conv_weights).clone().requires_grad_(True)
    nb = torch.as_tensor(neighbors).long()
    ends = torch.as_tensor(ends).long()
    m = ends.shape[0]
    rel = (t(pts_in)[nb[:, 1]] - t(pts_out)[nb[:, 0]]) * t(rho)
    basis = torch.nn.functional.gelu(rel @ a + b.reshape(1, -1))
    counts = torch.diff(ends, prepend=ends.new_zeros(1))
    seg = torch.repeat_interleave(torch.arange(m), counts)
    outer = x[nb[:, 1]].unsqueeze(2) * basis.unsqueeze(1)  # [E, C, K]
    tt = torch.zeros((m, x.shape[1], basis.shape[1]), dtype=dtype).index_add(0, seg, outer)
    out = torch.einsum("nik,iko->no", tt, w) * t(nu)
    out.backward(t(grad_out))
    return out.detach(), x.grad, a.grad, b.grad, w.grad



# ------------------------------------------------------------------- hierarchy build and frame pooling (rows f-2, f-3)
def grid_subsample(pts: torch.Tensor, batch_ids: torch.Tensor, cell_size: float):
    """One grid-average sub-sampling step: ``(cell_ids [N] int64, n_cells, level_pts [n_cells,3], level_batch)``.

    BoundingBox.py:17-18 (scatter_min - 1e-6 / scatter_max + 1e-6 per batch element), Grid.py:28-29 (cell counts =
    max over batches of int((max - min) / cell) + 1), Grid.py:37-45 (ComputeKeys with the same cell size in every
    dimension, then ``torch.unique(return_inverse=True)``: cells numbered in ascending key order),
    PointHierarchy.py:46-49 (level points = scatter_mean, level batch ids = scatter_max over the cells)."""
    pts = pts.to(torch.float32)
    b = batch_ids.to(torch.int64)
    nb = int(b.max()) + 1
    idx = b[:, None].expand(-1, 3)
    mx = torch.zeros((nb, 3)).scatter_reduce(0, idx, pts, "amax", include_self=False) + 1e-6
    mn = torch.zeros((nb, 3)).scatter_reduce(0, idx, pts, "amin", include_self=False) - 1e-6
    num_cells = torch.max(((mx - mn) / cell_size).to(torch.int32) + 1, dim=0)[0]
    keys = compute_keys(pts, batch_ids, mn, num_cells, torch.full((3,), cell_size, dtype=torch.float32))
    _, cell_ids = torch.unique(keys, return_inverse=True)
    n_cells = int(cell_ids.max()) + 1
    return cell_ids, n_cells, segment_pool(pts, cell_ids, n_cells, "avg"), segment_pool(batch_ids, cell_ids, n_cells, "max")


def segment_pool(x: torch.Tensor, cell_ids: torch.Tensor, n_cells: int, method: str) -> torch.Tensor:
    """``GridSubSample.__subsample_tensor__`` (GridSubSample.py:63-77): torch_scatter's ``scatter_mean`` (sum / count)
    and ``scatter_max`` over the cell ids; "min" / "sum" likewise (``scatter_min`` / ``scatter_add``)."""
    how = {"avg": "mean", "max": "amax", "min": "amin", "sum": "sum"}[method]
    idx = cell_ids.to(torch.int64)
    if x.dim() > 1:
        idx = idx.reshape((-1,) + (1,) * (x.dim() - 1)).expand_as(x)
    out = torch.zeros((n_cells,) + tuple(x.shape[1:]), dtype=x.dtype)
    if method in ("avg", "sum") or not x.requires_grad:
        return out.scatter_reduce(0, idx, x, how, include_self=False)
    # max / min with a gradient: torch_scatter routes it to ONE arg index per (cell, channel) (which one of several
    # equal values is not defined there); torch's own amax backward would split it between ties.  Take the first.
    ext = out.scatter_reduce(0, idx, x.detach(), how, include_self=False)
    rows = torch.arange(x.shape[0]).reshape((-1,) + (1,) * (x.dim() - 1)).expand_as(x)
    cand = torch.where(x.detach() == torch.gather(ext, 0, idx), rows, torch.full_like(rows, x.shape[0]))
    arg = torch.full(out.shape, x.shape[0], dtype=torch.int64).scatter_reduce(0, idx, cand, "amin", include_self=True)
    return torch.gather(x, 0, arg)


def segment_upsample(x: torch.Tensor, cell_ids: torch.Tensor) -> torch.Tensor:
    """``GridSubSample.__upsample_tensor__`` (GridSubSample.py:93): ``p_tensor[cell_ids]``."""
    return x[cell_ids.to(torch.int64)]


def frame_pool(x: torch.Tensor, n_frames: int, method: str) -> torch.Tensor:
    """``PointcloudRotEquiv.feature_pooling`` (PointcloudRotEquiv.py:224-251): scatter over the index
    ``repeat(arange(N), 'n -> (n t)', t=F)``, i.e. a reduction over the F consecutive rows of every point."""
    n = x.shape[0] // n_frames
    ids = torch.arange(n, dtype=torch.int64).repeat_interleave(n_frames)
    return segment_pool(x, ids, n, method)


# ------------------------------------------------------------ random grid sub-sample, two-cloud k-NN (rows f-2, f-1)
def grid_subsample_rnd(cell_ids: torch.Tensor, u: torch.Tensor):
    """``GridSubSample(..., p_rnd_sample=True)`` (GridSubSample.py:43-54) given the uniform numbers ``u [n_cells]`` the
    reference draws with ``torch.rand``: per-cell counts (scatter_add of ones over the sorted cell ids), exclusive
    offsets (cumsum, padded, :47-50), ``ids = floor(u * count) + offset`` (:52-54) -- positions in the cell-sorted
    point list ``sorted_ids = argsort(cell_ids)`` (Grid.py:48; stable here, the reference's argsort leaves the order
    inside a cell open).  Returns ``(sorted_ids, ids, picked = sorted_ids[ids])``; the product is clamped to
    count - 1 as the HIP path does (fp32 rounding of u * count)."""
    cid = cell_ids.to(torch.int64)
    sorted_ids = torch.argsort(cid, stable=True)
    counts = torch.bincount(cid)
    offsets = torch.cumsum(counts, 0) - counts
    off = torch.floor(u.to(torch.float32) * counts.to(torch.float32)).to(torch.int64)
    off = torch.minimum(off.clamp_min(0), counts - 1)
    ids = offsets + off
    return sorted_ids, ids, sorted_ids[ids]


def rows_upsample_rnd(x: torch.Tensor, picked: torch.Tensor, n_rows: int) -> torch.Tensor:
    """``__upsample_tensor__`` of the random mode (GridSubSample.py:83-91): zeros ``[n_rows, C]`` with the picked rows
    set to ``x`` (``scatter_`` along dim 0)."""
    out = torch.zeros((n_rows, x.shape[-1]), dtype=x.dtype)
    return out.scatter(0, picked.to(torch.int64)[:, None].expand(-1, x.shape[-1]), x)


def knn_query_pair(pts_src, batch_src, pts_q, batch_q, k: int) -> torch.Tensor:
    """k nearest SOURCE points of every query inside its batch element (the ``torch_cluster.knn(x=src, y=samples, k,
    batch_x, batch_y)`` call of KnnNeighborhood.py:77-84; torch-cluster 1.6.1 is not installed: documented semantics,
    parity unpinned).  ``[N_q, k]`` int32 source indices, ascending (squared distance, index), ``-1`` padded."""
    ps, pq = pts_src.to(torch.float32), pts_q.to(torch.float32)
    out = torch.full((pq.shape[0], k), -1, dtype=torch.int32)
    for b in torch.unique(batch_q):
        qi = torch.nonzero(batch_q == b)[:, 0]
        si = torch.nonzero(batch_src == b)[:, 0]
        if si.numel() == 0:
            continue
        d = pq[qi][:, None, :] - ps[si][None, :, :]
        d2 = torch.addcmul(torch.addcmul(d[..., 0] * d[..., 0], d[..., 1], d[..., 1]), d[..., 2], d[..., 2])
        order = torch.argsort(d2, dim=1, stable=True)[:, :k]
        out[qi, : order.shape[1]] = si[order].to(torch.int32)
    return out


# --------------------------------------------------- other relative-rotation descriptors (p_rel_rot, a1 / quirk 5)
def matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    """Rotation matrices ``[...,3,3]`` -> real-part-first quaternions (RotationFunctions.py:91-151, the pytorch3d
    formulation: the four candidate quaternions, the one with the largest denominator is taken)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(m.shape[:-2] + (9,)), dim=-1)
    q_abs = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(q_abs > 0, torch.sqrt(q_abs.clamp_min(0)), torch.zeros_like(q_abs))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return cand[best, :].reshape(m.shape[:-2] + (4,))


def edge_descriptors_rel(pts_in, pts_out, frames_in, frames_out, neighbors, rho, rel_rot: str = "6D") -> torch.Tensor:
    """``edge_descriptors`` for the three relative-rotation representations of ``get_relative_rot``
    (RotationFunctions.py:549-600): "6D" (9-D descriptor), "matrix" (all nine entries of R_out^T R_in: 12-D) and
    "quaternion" (7-D).  ``[E, F_out*F_in, D]``."""
    if rel_rot == "6D":
        return edge_descriptors(pts_in, pts_out, frames_in, frames_out, neighbors, rho)
    s = neighbors[:, 0].to(torch.int64)
    p = neighbors[:, 1].to(torch.int64)
    f_out, f_in = frames_out.shape[1], frames_in.shape[1]
    rel = (pts_in[p] - pts_out[s]) * rho
    r_out = frames_out[s].reshape(-1, f_out, 3, 3)
    r_in = frames_in[p].reshape(-1, f_in, 3, 3)
    loc = torch.matmul(rel[:, None, None, :], r_out).squeeze(2)[:, :, None, :].expand(-1, -1, f_in, -1)
    relrot = torch.matmul(r_out.transpose(2, 3)[:, :, None], r_in[:, None])  # [E,Fo,Fi,3,3]
    if rel_rot == "matrix":
        tail = relrot.reshape(-1, f_out, f_in, 9)
    elif rel_rot == "quaternion":
        tail = matrix_to_quaternion(relrot)
    else:
        raise ValueError(rel_rot)
    return torch.cat((loc, tail), -1).reshape(-1, f_out * f_in, 3 + tail.shape[-1])
