#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by running the REFERENCE's own Python.

Run only in the build container (``/root/reference`` does not exist on the GPU box):

    python tools/gen_golden.py

What is executed from the reference (imported from /root/reference/point_cloud_lib, never
copied): ``PNEConvLayerRotEquivFactory`` / ``PNEConvLayerRotEquiv`` (forward through
``IConvLayer.forward`` incl. the EMA pre-process branch, ``get_rot_tenors``, the kernel MLP,
``FeatBasisProj`` autograd function, einsum, scalings, parameter init), ``PointcloudRotEquiv``
(random frames), ``BQNeighborhood``, and ``RotationFunctions.*``.

What is NOT the reference: three native modules are not installed here and cannot be built
(no nvcc): ``point_cloud_lib_ops`` (CUDA), ``torch_scatter``, ``torch_cluster``.  The
stand-ins below are deliberately naive loop/index_add implementations of their documented
semantics, written independently of ``oracle/se3conv_oracle.py`` so the fixtures cross-check
the oracle instead of echoing it.  Everything stored is data (inputs + outputs), no source.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/point_cloud_lib"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ----------------------------------------------------------------------------- stand-ins
def _install_shims():
    ops = types.ModuleType("point_cloud_lib_ops")

    def feat_basis_proj(basis, feats, neighbors, ends):
        m, c, k = ends.shape[0], feats.shape[1], basis.shape[1]
        out = torch.zeros((m, c, k), dtype=feats.dtype)
        start = 0
        for row in range(m):
            end = int(ends[row])
            if end > start:
                f = feats[neighbors[start:end, 1].long()]  # [n,C]
                out[row] = f.t() @ basis[start:end]
            start = end
        return out

    def feat_basis_proj_grad(basis, feats, neighbors, ends, grads):
        g_feat = torch.zeros_like(feats)
        g_basis = torch.zeros_like(basis)
        start = 0
        for row in range(ends.shape[0]):
            end = int(ends[row])
            if end > start:
                src = neighbors[start:end, 1].long()
                g = grads[row]  # [C,K]
                g_basis[start:end] = feats[src] @ g
                g_feat.index_add_(0, src, basis[start:end] @ g.t())
            start = end
        return [g_feat, g_basis]

    def ball_query(src, dst, bsrc, bdst, min_pt, num_cells, radius, max_neighbors):
        assert max_neighbors == 0
        inv = torch.reciprocal(radius)
        nbrs, ends, total = [], [], 0
        for s in range(dst.shape[0]):
            d = (dst[s][None, :] - src) * inv[None, :]
            dist = torch.sqrt((d * d).sum(1))
            hit = torch.nonzero((dist < 1.0) & (bsrc == bdst[s]))[:, 0]
            nbrs.append(torch.stack((torch.full_like(hit, s), hit), 1))
            total += hit.shape[0]
            ends.append(total)
        return torch.cat(nbrs).long(), torch.tensor(ends, dtype=torch.int32)

    def _unavailable(*a, **k):
        raise RuntimeError("native op not available in the fixture generator")

    ops.feat_basis_proj = feat_basis_proj
    ops.feat_basis_proj_grad = feat_basis_proj_grad
    ops.ball_query = ball_query
    def knn_query(pts, batch_ids, k):
        n = pts.shape[0]
        out = torch.full((n, k), -1, dtype=torch.int32)
        for i in range(n):
            same = torch.nonzero(batch_ids == batch_ids[i])[:, 0]
            d = pts[same] - pts[i][None, :]
            d2 = (d * d).sum(1)
            order = torch.argsort(d2, stable=True)[:k]
            out[i, : order.shape[0]] = same[order].to(torch.int32)
        return out

    ops.knn_query = knn_query
    def compute_keys(pts, batch_ids, aabb_min, num_cells, cell_size):
        # compute_keys.cu:32-71, grid_utils.cuh:57-93: cell = clamp(floor((p - min[b]) * (1/cell))), row-major key
        inv = torch.reciprocal(cell_size.to(torch.float32))
        rel = (pts.to(torch.float32) - aabb_min[batch_ids.long()]) * inv
        nc = num_cells.to(torch.int64)
        cell = torch.minimum(torch.maximum(torch.floor(rel).to(torch.int64), torch.zeros(3, dtype=torch.int64)), nc - 1)
        return ((batch_ids.to(torch.int64) * nc[0] + cell[:, 0]) * nc[1] + cell[:, 1]) * nc[2] + cell[:, 2]

    ops.compute_keys = compute_keys
    sys.modules["point_cloud_lib_ops"] = ops

    ts = types.ModuleType("torch_scatter")

    def _size(index, dim_size):
        return int(index.max()) + 1 if dim_size is None else dim_size

    def scatter_add(src, index, dim=0, out=None, dim_size=None):
        assert dim == 0
        n = _size(index, dim_size)
        res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
        return res.index_add_(0, index.long(), src)

    def _reduce(src, index, how, dim_size):
        n = _size(index, dim_size)
        idx = index.long()
        if src.dim() > 1:
            idx = idx[:, None].expand_as(src)
        init = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
        return init.scatter_reduce(0, idx, src, how, include_self=False)

    ts.scatter_add = scatter_add
    ts.scatter_mean = lambda src, index, dim=0, out=None, dim_size=None: _reduce(src, index, "mean", dim_size)
    ts.scatter_max = lambda src, index, dim=0, out=None, dim_size=None: (_reduce(src, index, "amax", dim_size), None)
    ts.scatter_min = lambda src, index, dim=0, out=None, dim_size=None: (_reduce(src, index, "amin", dim_size), None)
    sys.modules["torch_scatter"] = ts

    tc = types.ModuleType("torch_cluster")
    for name in ("knn", "knn_graph", "fps", "radius"):
        setattr(tc, name, _unavailable)
    sys.modules["torch_cluster"] = tc
    sys.modules["h5py"] = types.ModuleType("h5py")
    # the reference's data_sets package imports a file that is not in the repository
    sys.modules["point_cloud_lib.data_sets"] = types.ModuleType("point_cloud_lib.data_sets")
    for name in ("wandb", "trimesh", "webdataset"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)


def _import_reference():
    _install_shims()
    import point_cloud_lib as pclib  # noqa: E402

    return pclib


# ----------------------------------------------------------------------------- cases
def _radius(n, k):
    return float((3.0 * k / (4.0 * np.pi * n)) ** (1.0 / 3.0))


def layer_case(pclib, seed, n_in, n_out, f, c_in, c_out, k_deg, num_basis=32, batches=1):
    """One conv between an input cloud and an output cloud (same cloud when n_out is None)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    cfg = {"pca": False, "n_frames": f, "fixed_axis": False}
    pts_in = torch.rand(n_in, 3)
    bid_in = torch.sort(torch.randint(0, batches, (n_in,), dtype=torch.int32)).values
    pc_in = pclib.pc.PointcloudRotEquiv(pts_in, bid_in, cfg)
    if n_out is None:
        pc_out = pc_in
    else:
        pts_out = torch.rand(n_out, 3)
        bid_out = torch.sort(torch.randint(0, batches, (n_out,), dtype=torch.int32)).values
        pc_out = pclib.pc.PointcloudRotEquiv(pts_out, bid_out, cfg)
    r = _radius(n_in / batches, k_deg)
    neigh = pclib.pc.BQNeighborhood(pc_in, pc_out, r)

    factory = pclib.layers.PNEConvLayerRotEquivFactory(9, num_basis, "mlp_gelu")
    conv = factory.create_conv_layer(c_in, c_out)
    with torch.no_grad():
        conv.proj_biases_.uniform_(-0.5, 0.5)  # the init is 0; make the bias path visible

    x = torch.randn(n_in * f, c_in)
    ema = []
    conv.start_pre_process()
    with torch.no_grad():
        for _ in range(3):
            pclib.layers.PNEConvLayerRotEquiv.empty_rot_tenors_cache()
            conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=x, p_neighborhood=neigh)
            ema.append((float(conv.norm_neigh_dist_), float(conv.norm_num_neighs_)))
    conv.end_pre_process()
    # use converged normalisers for the numeric case (the EMA trajectory is stored separately)
    conv.norm_neigh_dist_ = torch.tensor(1.0 / r, dtype=torch.float32)
    conv.norm_num_neighs_ = torch.tensor(neigh.start_ids_.shape[0] / neigh.neighbors_.shape[0], dtype=torch.float32)

    pclib.layers.PNEConvLayerRotEquiv.empty_rot_tenors_cache()
    x.requires_grad_(True)
    out = conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=x, p_neighborhood=neigh)
    g = torch.randn_like(out)
    out.backward(g)
    rt = pclib.layers.PNEConvLayerRotEquiv.get_rot_tenors(pc_in, pc_out, neigh, conv.norm_neigh_dist_)

    return {
        "pts_in": pc_in.pts_.numpy(), "pts_out": pc_out.pts_.numpy(),
        "batch_in": pc_in.batch_ids_.numpy(), "batch_out": pc_out.batch_ids_.numpy(),
        "frames_in": pc_in.local_frames_.numpy(), "frames_out": pc_out.local_frames_.numpy(),
        "radius": np.float64(r),
        "neighbors": neigh.neighbors_.numpy().astype(np.int32), "ends": neigh.start_ids_.numpy().astype(np.int32),
        "proj_axes": conv.proj_axes_.detach().numpy(), "proj_biases": conv.proj_biases_.detach().numpy(),
        "conv_weights": conv.conv_weights_.detach().numpy(),
        "rho": conv.norm_neigh_dist_.numpy(), "nu": conv.norm_num_neighs_.numpy(),
        "x": x.detach().numpy(), "out": out.detach().numpy(), "grad_out": g.numpy(),
        "dx": x.grad.numpy(), "dA": conv.proj_axes_.grad.numpy(), "dbeta": conv.proj_biases_.grad.numpy(),
        "dW": conv.conv_weights_.grad.numpy(),
        "rt_desc": rt["rel_pts_rel_orient"].numpy(), "rt_neighbs": rt["neighbs"].numpy().astype(np.int32),
        "rt_ends": rt["neighbs_start_ids"].numpy().astype(np.int32),
        "ema": np.asarray(ema, dtype=np.float64),
    }


def rotation_case(pclib, seed):
    """get_relative_rot / change_direction_to_local_frame / all_index_combinations on tiny inputs."""
    torch.manual_seed(seed)
    fa = pclib.pc.sample_reference_frames(5, 2)
    fb = pclib.pc.sample_reference_frames(5, 4)
    d = torch.randn(5, 3)
    return {
        "frames_a": fa.numpy(), "frames_b": fb.numpy(), "dirs": d.numpy(),
        "rel6": pclib.pc.get_relative_rot(fa, fb, "6D").numpy(),
        "relmat": pclib.pc.get_relative_rot(fa, fb, "matrix").numpy(),
        "local": pclib.pc.change_direction_to_local_frame(d, fa).numpy(),
        "combos": pclib.pc.all_index_combinations(2, 4).numpy(),
    }


def pca_case(pclib, seed):
    """PCA frames through the reference's PointcloudRotEquiv (pca: True, knn 16): the un-shuffled 'se3-all'
    cache for the free case and for a fixed up-axis, plus the kNN ids the (stand-in) kNN produced."""
    torch.manual_seed(seed)
    n = 240
    pts = torch.rand(n, 3) * torch.tensor([1.0, 0.7, 0.4])
    bid = torch.sort(torch.randint(0, 2, (n,), dtype=torch.int32)).values
    out = {"pts": pts.numpy(), "batch": bid.numpy()}
    for tag, axis in (("free", False), ("axis2", 2), ("axis1", 1)):
        cfg = {"pca": True, "n_frames": 2, "fixed_axis": axis, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
        pc = pclib.pc.PointcloudRotEquiv(pts, bid, cfg)
        out[f"frames_{tag}"] = pc.local_frames_pca_cache_["se3-all"].numpy()
        nbh = pc.get_ref_frame_neighborhood("knn", neigh_k=16)
        out["knn"] = nbh.neighbors_[:, 1].reshape(n, 16).numpy().astype(np.int32)
        assert pc.local_frames_.shape == (n, 2, 9)
    return out


def hierarchy_case(pclib, seed):
    """Scope rows f-2 / f-3 through the reference's Python: PointHierarchy (grid_avg, two sub-samples: Grid,
    BoundingBox, GridSubSample with the scatter stand-ins), pool_tensor avg / max with gradients, upsample_tensor with
    gradient, and PointcloudRotEquiv.feature_pooling for every method with gradients."""
    torch.manual_seed(seed)
    n, c, f = 700, 12, 4
    pts = torch.rand(n, 3) * torch.tensor([1.0, 0.8, 0.5])
    bid = torch.sort(torch.randint(0, 3, (n,), dtype=torch.int32)).values
    cells = [0.11, 0.23]
    hier = pclib.pc.PointHierarchy(pclib.pc.Pointcloud(pts, bid), 2, "grid_avg", grid_radii=cells)
    out = {"pts": pts.numpy(), "batch": bid.numpy(), "cells": np.array(cells, dtype=np.float32)}
    for lv in (1, 2):
        out[f"pts_l{lv}"] = hier.pcs_[lv].pts_.numpy()
        out[f"batch_l{lv}"] = hier.pcs_[lv].batch_ids_.numpy().astype(np.int32)
        out[f"cell_ids_l{lv - 1}"] = hier.sub_sampled_objs_[lv - 1].grid_.cell_ids_.numpy().astype(np.int32)
    for method in ("avg", "max"):
        x = torch.randn(n, c, requires_grad=True)
        y = hier.pool_tensor(x, 0, 1, method)
        g = torch.randn_like(y)
        y.backward(g)
        out[f"pool_{method}_x"], out[f"pool_{method}_y"] = x.detach().numpy(), y.detach().numpy()
        out[f"pool_{method}_g"], out[f"pool_{method}_dx"] = g.numpy(), x.grad.numpy()
    z = torch.randn(hier.pcs_[1].pts_.shape[0], c, requires_grad=True)
    up = hier.upsample_tensor(z, 1, 0)
    gu = torch.randn_like(up)
    up.backward(gu)
    out["up_z"], out["up_y"], out["up_g"], out["up_dz"] = z.detach().numpy(), up.detach().numpy(), gu.numpy(), z.grad.numpy()
    cfg = {"pca": False, "n_frames": f, "fixed_axis": False}
    pcr = pclib.pc.PointcloudRotEquiv(pts[:150], bid[:150], cfg)
    out["frames"] = np.int32(f)
    for method in ("avg", "max", "min", "sum"):
        xf = torch.randn(150 * f, c, requires_grad=True)
        yf = pcr.feature_pooling(xf, method)
        gf = torch.randn_like(yf)
        yf.backward(gf)
        out[f"fpool_{method}_x"], out[f"fpool_{method}_y"] = xf.detach().numpy(), yf.detach().numpy()
        out[f"fpool_{method}_g"], out[f"fpool_{method}_dx"] = gf.numpy(), xf.grad.numpy()
    for method in ("avg", "max"):
        xg = torch.randn(150 * f, c)
        out[f"gpool_{method}_x"] = xg.numpy()
        out[f"gpool_{method}_y"] = pcr.global_pooling(xg, method).numpy()
        out[f"gpool2_{method}_y"] = pcr.global_pooling_specific_feature_pooling(xg, method, "max").numpy()
    out["gpool_batch"] = bid[:150].numpy()
    return out


def grid_rnd_case(pclib, seed):
    """GridSubSample(..., p_rnd_sample=True) of the reference (pc/GridSubSample.py:43-54, 66-67, 83-91) as the task
    scripts use it for the output cloud (tasks/SemSeg/train_dfaust_rot.py:143-149), plus a PointHierarchy built with
    "grid_rnd".  Stored: the uniform numbers the reference drew (its first RNG call after the seed: re-drawn here and
    checked against `ids_`), `ids_`, `sorted_ids_`, `cell_ids_`, the selected points, the sub-sampled points / batch ids /
    labels, an up-sampled feature tensor and the gradients of both maps."""
    n, c = 600, 10
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(n, 3, generator=g) * torch.tensor([1.0, 0.8, 0.5])
    bid = torch.sort(torch.randint(0, 3, (n,), dtype=torch.int32, generator=g)).values
    labels = torch.randint(0, 7, (n,), dtype=torch.int64, generator=g)
    cell = 0.13
    pc = pclib.pc.Pointcloud(pts, bid)
    torch.manual_seed(seed + 100)
    samp = pclib.pc.GridSubSample(pc, cell, p_rnd_sample=True)
    n_cells = int(samp.grid_.cell_ids_.max()) + 1
    torch.manual_seed(seed + 100)
    u = torch.rand(n_cells)  # the numbers GridSubSample.py:52 drew
    counts = torch.bincount(samp.grid_.cell_ids_.long())
    assert torch.equal(torch.floor(u * counts).to(torch.int32) + (torch.cumsum(counts, 0) - counts).to(torch.int32), samp.ids_)
    picked = samp.grid_.sorted_ids_[samp.ids_.long()]
    out = {"pts": pts.numpy(), "batch": bid.numpy(), "labels": labels.numpy(), "cell": np.float32(cell), "u": u.numpy(),
           "ids": samp.ids_.numpy().astype(np.int32), "sorted_ids": samp.grid_.sorted_ids_.numpy().astype(np.int32),
           "cell_ids": samp.grid_.cell_ids_.numpy().astype(np.int32), "picked": picked.numpy().astype(np.int32),
           "sub_pts": samp.__subsample_tensor__(pc.pts_, "avg").numpy(),
           "sub_batch": samp.__subsample_tensor__(pc.batch_ids_, "max").numpy().astype(np.int32),
           "sub_labels": samp.__subsample_tensor__(labels, "max").numpy()}
    x = torch.randn(n, c, generator=g, requires_grad=True)
    y = samp.__subsample_tensor__(x, "avg")
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out["x"], out["sub_x"], out["sub_g"], out["sub_dx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    z = torch.randn(n_cells, c, generator=g, requires_grad=True)
    up = samp.__upsample_tensor__(z)
    gu = torch.randn(up.shape, generator=g)
    up.backward(gu)
    out["z"], out["up_y"], out["up_g"], out["up_dz"] = z.detach().numpy(), up.detach().numpy(), gu.numpy(), z.grad.numpy()
    # a two-step hierarchy with "grid_rnd": the random numbers of each step, the level points / batch ids
    cells = [0.11, 0.23]
    torch.manual_seed(seed + 200)
    hier = pclib.pc.PointHierarchy(pclib.pc.Pointcloud(pts, bid), 2, "grid_rnd", grid_radii=cells)
    torch.manual_seed(seed + 200)
    out["h_cells"] = np.array(cells, dtype=np.float32)
    for lv in (1, 2):
        nc = hier.pcs_[lv].pts_.shape[0]
        out[f"h_u{lv}"] = torch.rand(nc).numpy()
        out[f"h_pts{lv}"] = hier.pcs_[lv].pts_.numpy()
        out[f"h_batch{lv}"] = hier.pcs_[lv].batch_ids_.numpy().astype(np.int32)
        out[f"h_picked{lv}"] = hier.sub_sampled_objs_[lv - 1].grid_.sorted_ids_[hier.sub_sampled_objs_[lv - 1].ids_.long()].numpy().astype(np.int32)
    return out


def rel_rot_case(pclib, seed, rel_rot, dims):
    """PNEConvLayerRotEquiv with p_rel_rot = 'matrix' (12-D descriptor) / 'quaternion' (7-D): forward, backward and the
    materialised rot tensors through the reference's Python (PNEConvLayerRotEquiv.py:236-281, RotationFunctions.py:593-600)."""
    torch.manual_seed(seed)
    n, f, c_in, c_out = 160, 2, 8, 16
    pts = torch.rand(n, 3)
    bid = torch.zeros(n, dtype=torch.int32)
    pc = pclib.pc.PointcloudRotEquiv(pts, bid, {"pca": False, "n_frames": f, "fixed_axis": False})
    r = _radius(n, 10)
    neigh = pclib.pc.BQNeighborhood(pc, pc, r)
    conv = pclib.layers.PNEConvLayerRotEquivFactory(dims, 32, "mlp_gelu", rel_rot).create_conv_layer(c_in, c_out)
    with torch.no_grad():
        conv.proj_biases_.uniform_(-0.5, 0.5)
    conv.norm_neigh_dist_ = torch.tensor(1.0 / r, dtype=torch.float32)
    conv.norm_num_neighs_ = torch.tensor(neigh.start_ids_.shape[0] / neigh.neighbors_.shape[0], dtype=torch.float32)
    pclib.layers.PNEConvLayerRotEquiv.empty_rot_tenors_cache()
    x = torch.randn(n * f, c_in, requires_grad=True)
    out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=neigh)
    g = torch.randn_like(out)
    out.backward(g)
    rt = pclib.layers.PNEConvLayerRotEquiv.get_rot_tenors(pc, pc, neigh, conv.norm_neigh_dist_)
    pclib.layers.PNEConvLayerRotEquiv.rel_rot_type = "6D"  # class attribute set by the last-created factory: restore
    pclib.layers.PNEConvLayerRotEquiv.empty_rot_tenors_cache()
    return {"pts": pts.numpy(), "batch": bid.numpy(), "frames": pc.local_frames_.numpy(), "radius": np.float64(r),
            "neighbors": neigh.neighbors_.numpy().astype(np.int32), "ends": neigh.start_ids_.numpy().astype(np.int32),
            "proj_axes": conv.proj_axes_.detach().numpy(), "proj_biases": conv.proj_biases_.detach().numpy(),
            "conv_weights": conv.conv_weights_.detach().numpy(), "rho": conv.norm_neigh_dist_.numpy(),
            "nu": conv.norm_num_neighs_.numpy(), "x": x.detach().numpy(), "out": out.detach().numpy(), "grad_out": g.numpy(),
            "dx": x.grad.numpy(), "dA": conv.proj_axes_.grad.numpy(), "dbeta": conv.proj_biases_.grad.numpy(),
            "dW": conv.conv_weights_.grad.numpy(), "rt_desc": rt["rel_pts_rel_orient"].numpy(),
            "rt_neighbs": rt["neighbs"].numpy().astype(np.int32), "rt_ends": rt["neighbs_start_ids"].numpy().astype(np.int32)}


def state_dict_keys(pclib):
    """Parameter / buffer names and shapes of the reference's modules around the hot path (checkpoint layout): the two
    convolutions, BatchNormPC, SkipConnection and a whole ResNetFormer block.  Names and shapes only -- no values."""
    rot = pclib.layers.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
    pne = pclib.layers.PNEConvLayerFactory(3, 32, "mlp_gelu")
    mods = {
        "PNEConvLayerRotEquiv(9,16,24,32)": rot.create_conv_layer(16, 24),
        "PNEConvLayer(3,16,24,32)": pne.create_conv_layer(16, 24),
        "BatchNormPC(24)": pclib.layers.BatchNormPC(24),
        "SkipConnection(24)": pclib.layers.SkipConnection(0.1, 24),
        "ResNetFormer(16,24)": pclib.layers.ResNetFormer(16, 24, rot, pclib.layers.BatchNormPC, 0.1),
        "ResNetFormer(24,24)": pclib.layers.ResNetFormer(24, 24, rot, pclib.layers.BatchNormPC, 0.0),
    }
    return {name: {k: list(v.shape) for k, v in m.state_dict().items()} for name, m in mods.items()}


def block_case(pclib, seed):
    """A whole ResNetFormer block of the reference (BatchNormPC in training mode, SkipConnection with gamma moved off
    its 1e-6 initial value, no drop path) around PNEConvLayerRotEquiv: state_dict, input, output and every gradient."""
    torch.manual_seed(seed)
    n, f, c_in, c_out = 220, 2, 32, 48
    pts = torch.rand(n, 3)
    bid = torch.sort(torch.randint(0, 2, (n,), dtype=torch.int32)).values
    pc = pclib.pc.PointcloudRotEquiv(pts, bid, {"pca": False, "n_frames": f, "fixed_axis": False})
    r = _radius(n / 2, 12)
    nbh = pclib.pc.BQNeighborhood(pc, pc, r)
    fact = pclib.layers.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
    blk = pclib.layers.ResNetFormer(c_in, c_out, fact, pclib.layers.BatchNormPC, 0.0)
    blk.train()
    with torch.no_grad():
        blk.spatial_conv_.norm_neigh_dist_.fill_(1.0 / r)
        blk.spatial_conv_.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / nbh.neighbors_.shape[0])
        blk.spatial_conv_.proj_biases_.uniform_(-0.5, 0.5)
        blk.skip_path_1_.gamma_.fill_(0.7)
        blk.skip_path_2_.gamma_.fill_(1.3)
    state0 = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    x = torch.randn(n * f, c_in, requires_grad=True)
    out = blk(pc, x, nbh)
    g = torch.randn_like(out)
    out.backward(g)
    data = {"pts": pts.numpy(), "batch": bid.numpy(), "frames": pc.local_frames_.numpy(), "radius": np.float32(r),
            "x": x.detach().numpy(), "out": out.detach().numpy(), "g": g.numpy(), "dx": x.grad.numpy()}
    for k, v in state0.items():
        data["state/" + k] = v.numpy()
    for k, v in blk.named_parameters():
        data["grad/" + k] = v.grad.numpy()
    for k, v in blk.state_dict().items():
        if "running" in k:
            data["after/" + k] = v.numpy()
    return data


def pne_case(pclib, seed, n_in, n_out, c_in, c_out, k_deg, batches):
    """The reference's non-equivariant PNEConvLayer ('mlp_gelu', aggregation 'add', 3-D offsets; scope row f-4):
    plain Pointcloud, BQNeighborhood, forward + backward through LinearPNE / FeatBasisProj / einsum."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    pts_in = torch.rand(n_in, 3)
    bid_in = torch.sort(torch.randint(0, batches, (n_in,), dtype=torch.int32)).values
    pc_in = pclib.pc.Pointcloud(pts_in, bid_in)
    if n_out is None:
        pc_out = pc_in
    else:
        pts_out = torch.rand(n_out, 3)
        bid_out = torch.sort(torch.randint(0, batches, (n_out,), dtype=torch.int32)).values
        pc_out = pclib.pc.Pointcloud(pts_out, bid_out)
    r = _radius(n_in / batches, k_deg)
    neigh = pclib.pc.BQNeighborhood(pc_in, pc_out, r)
    conv = pclib.layers.PNEConvLayerFactory(3, 32, "mlp_gelu").create_conv_layer(c_in, c_out)
    with torch.no_grad():
        conv.proj_biases_.uniform_(-0.5, 0.5)
    conv.norm_neigh_dist_ = torch.tensor(1.0 / r, dtype=torch.float32)
    conv.norm_num_neighs_ = torch.tensor(neigh.start_ids_.shape[0] / neigh.neighbors_.shape[0], dtype=torch.float32)
    x = torch.randn(n_in, c_in, requires_grad=True)
    out = conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=x, p_neighborhood=neigh)
    g = torch.randn_like(out)
    out.backward(g)
    return {
        "pts_in": pc_in.pts_.numpy(), "pts_out": pc_out.pts_.numpy(),
        "batch_in": pc_in.batch_ids_.numpy(), "batch_out": pc_out.batch_ids_.numpy(), "radius": np.float64(r),
        "neighbors": neigh.neighbors_.numpy().astype(np.int32), "ends": neigh.start_ids_.numpy().astype(np.int32),
        "proj_axes": conv.proj_axes_.detach().numpy(), "proj_biases": conv.proj_biases_.detach().numpy(),
        "conv_weights": conv.conv_weights_.detach().numpy(),
        "rho": conv.norm_neigh_dist_.numpy(), "nu": conv.norm_num_neighs_.numpy(),
        "x": x.detach().numpy(), "out": out.detach().numpy(), "grad_out": g.numpy(),
        "dx": x.grad.numpy(), "dA": conv.proj_axes_.grad.numpy(), "dbeta": conv.proj_biases_.grad.numpy(),
        "dW": conv.conv_weights_.grad.numpy(),
    }



sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from seeded_params import seeded_conv_params as _seeded_conv_params  # noqa: E402  (shared with the tests that replay the fixtures)

NET_SAMPLE = 2048  # entries of a weight gradient with more than 4 096 entries that the fixture keeps (seeded positions)
OUT_SAMPLE = 2048  # "scannet": entries of an output / feature gradient with more than 8 192 entries that the fixture keeps


def _faust_bodies(seed):
    """Two synthetic bodies (thin shells around ellipsoids of roughly a torso's proportions), one constant input channel."""
    def body(n, scale, g):
        u = torch.randn(n, 3, generator=g)
        u = u / u.norm(dim=1, keepdim=True)
        return u * torch.tensor([0.25, 0.85, 0.15]) * scale + 0.01 * torch.randn(n, 3, generator=g)

    g = torch.Generator().manual_seed(seed + 1)
    n_raw = 1100
    pts = torch.cat([body(n_raw, 0.40, g), body(n_raw, 0.34, g) + torch.tensor([2.0, 0.0, 0.0])])
    bid = torch.cat([torch.zeros(n_raw, dtype=torch.int32), torch.ones(n_raw, dtype=torch.int32)])
    return pts, bid, torch.ones(2 * n_raw, 1), g


def _scannet_rooms(seed):
    """Two synthetic rooms: floor, two walls and a box of furniture each, points on the surfaces with scanner noise; six
    input columns (position copy + colour) as the ScanNet loader hands over, of which create_hierarchy keeps the last three
    (tasks/SemSeg/train_scannet_rot.py:151-156)."""
    g = torch.Generator().manual_seed(seed + 1)

    def plane(n, origin, u, v):
        ab = torch.rand(n, 2, generator=g)
        return torch.tensor(origin) + ab[:, :1] * torch.tensor(u) + ab[:, 1:] * torch.tensor(v)

    def room(w, d, hgt, n):
        parts = [plane(n, [0., 0., 0.], [w, 0., 0.], [0., d, 0.]),              # floor
                 plane(n // 2, [0., 0., 0.], [w, 0., 0.], [0., 0., hgt]),        # wall y = 0
                 plane(n // 2, [0., 0., 0.], [0., d, 0.], [0., 0., hgt]),        # wall x = 0
                 plane(n // 4, [0.4, 0.3, 0.3], [0.4, 0., 0.], [0., 0.3, 0.]),   # table top
                 plane(n // 6, [0.4, 0.3, 0.], [0.4, 0., 0.], [0., 0., 0.3])]    # table front
        p = torch.cat(parts)
        return p + 0.004 * torch.randn(p.shape, generator=g)

    r0, r1 = room(1.1, 0.9, 0.6, 1100), room(0.9, 1.0, 0.55, 1000)
    pts = torch.cat([r0, r1 + torch.tensor([6.0, 1.0, 0.0])])
    bid = torch.cat([torch.zeros(r0.shape[0], dtype=torch.int32), torch.ones(r1.shape[0], dtype=torch.int32)])
    colour = torch.rand(pts.shape[0], 3, generator=g)
    return pts, bid, torch.cat([pts, colour], 1), g


NETWORKS = {
    # BASELINE.json config 2: confs/dfaust/dfaust_I_rot_pca_2F.yaml
    "faust": dict(model="FPNSegUNetMLPGeluRotEqFAUST", ctor=(1, 8, 0.5, 0.0), n_convs=21, param_base=7000, scene=_faust_bodies,
                  drop_cols=0,
                  md={"init_subsample": 0.04, "output_subsample": 0.04, "grid_subsamples": [0.05, 0.1, 0.2, 0.4],
                      "RefFrames": {"pca": True, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}, "fixed_axis": False,
                                    "n_frames": 2}}),
    # BASELINE.json config 3: confs/scannet/scannet20_rot_pca_SO2.yaml:26-41 (init 0.1, grids 0.2 ... 1.6, PCA frames about the
    # fixed axis 2, train_n_frames = 1), tasks/SemSeg/seg_models.py:39-58,90-95 (blocks [2,3,4,6,4], widths [64 ... 320], FPN 128)
    "scannet": dict(model="FPNSegUNetMLPGeluRotEqScanNet", ctor=(3, 20, 0.5), n_convs=32, param_base=7100, scene=_scannet_rooms,
                    drop_cols=3,
                    md={"init_subsample": 0.1, "output_subsample": 0.1, "grid_subsamples": [0.2, 0.4, 0.8, 1.6],
                        "RefFrames": {"pca": True, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}, "fixed_axis": 2,
                                      "n_frames": 1}}),
}


def network_case(pclib, seed, net="faust"):
    """BASELINE.json config 2 (net = "faust") / config 3 (net = "scannet", the reference's FPNSegUNetMLPGeluRotEqScanNet fed as
    tasks/SemSeg/train_scannet_rot.py:142-186,262-290 does: 32 convolution calls, F = 1 frames about the up axis).  FAUST: the reference's own FPNSegUNetMLPGeluRotEqFAUST (models/FPNSegUNet.py:198-223,
    Encoder.py:116-173, FPNDecoder.py:87-137, tasks/SemSeg/seg_models.py:16-108) built by the reference, fed through the
    call sequence of the task script (create_hierarchy, tasks/SemSeg/train_dfaust_rot.py:108-158, with the configuration
    confs/dfaust/dfaust_I_rot_pca_2F.yaml: init / output sub-sample 0.04, grid sub-samples 0.05 ... 0.4, PCA frames from
    kNN 16, F = 2) on two synthetic bodies, one pre-process pass (EMA buffers) and one training-mode forward + backward.
    A forward hook on every PNEConvLayerRotEquiv records what the network feeds to and gets from each of its 21
    convolution calls: clouds, neighbourhood, buffers, input, output, and the gradients autograd delivers (tensor hooks).
    Weights are re-drawn per conv from a seed (see _seeded_conv_params) before the pass; weight gradients above 4 096
    entries are kept at NET_SAMPLE seeded positions plus their norm.  Data only."""
    import importlib

    from einops import repeat

    for path in ("/root/reference", "/root/reference/tasks/SemSeg"):
        if path not in sys.path:
            sys.path.insert(0, path)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        seg = importlib.import_module("seg_models")
    torch.manual_seed(seed)
    np.random.seed(seed)
    cfg = NETWORKS[net]
    model = getattr(seg, cfg["model"])(*cfg["ctor"])
    convs = [m for m in model.modules() if isinstance(m, pclib.layers.PNEConvLayerRotEquiv)]
    md = cfg["md"]
    pts, bid, feats, g = cfg["scene"](seed)
    with torch.no_grad():  # create_hierarchy(p_init_subsample=True) of the task script
        pc = pclib.pc.Pointcloud(pts, bid)
        samp = pclib.pc.GridSubSample(pc, md["init_subsample"])
        new_pts = samp.__subsample_tensor__(pc.pts_, "avg")
        new_bid = samp.__subsample_tensor__(pc.batch_ids_, "max")
        new_f = samp.__subsample_tensor__(feats, "avg")[:, cfg["drop_cols"]:]
        new_pc = pclib.pc.PointcloudRotEquiv(new_pts, new_bid, md["RefFrames"])
        hier = pclib.pc.PointHierarchyRotEquiv(new_pc, len(md["grid_subsamples"]), "grid_avg", grid_radii=md["grid_subsamples"])
        radii = [md["init_subsample"]] + md["grid_subsamples"]
        samp2 = pclib.pc.GridSubSample(pc, md["output_subsample"], p_rnd_sample=True)
        out_pc = pclib.pc.PointcloudRotEquiv(samp2.__subsample_tensor__(pc.pts_, "avg"),
                                             samp2.__subsample_tensor__(pc.batch_ids_, "max"), md["RefFrames"])
    x0 = repeat(new_f, "n d -> (n times) d", times=md["RefFrames"]["n_frames"])
    # seeded parameters (see _seeded_conv_params), in the order model.modules() lists the convolutions
    with torch.no_grad():
        for i, c in enumerate(convs):
            a, b, w = _seeded_conv_params(i, 9, c.conv_weights_.shape[0], 32, c.conv_weights_.shape[2], cfg["param_base"])
            c.proj_axes_.copy_(a), c.proj_biases_.copy_(b), c.conv_weights_.copy_(w)
    model.eval()
    model.start_pre_process()
    with torch.no_grad():
        for _ in range(3):
            model(hier, x0, radii, out_pc)
    model.end_pre_process()
    model.train()

    calls = []

    def hook(mod, args, kwargs, out):
        rec = {"mod": mod, "kw": kwargs, "out": out, "grad_out": None, "dx": None}
        out.register_hook(lambda gr, rec=rec: rec.__setitem__("grad_out", gr.detach().clone()))
        xin = kwargs["p_in_features"]
        if xin.requires_grad:
            xin.register_hook(lambda gr, rec=rec: rec.__setitem__("dx", gr.detach().clone()))
        calls.append(rec)

    handles = [c.register_forward_hook(hook, with_kwargs=True) for c in convs]
    y = model(hier, x0, radii, out_pc)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    for h in handles:
        h.remove()
    assert len(calls) == len(convs) == cfg["n_convs"], (len(calls), len(convs))

    clouds, nbhs = [], []

    def cloud_id(pcx):
        for i, c in enumerate(clouds):
            if c is pcx:
                return i
        clouds.append(pcx)
        return len(clouds) - 1

    def nbh_id(nb):
        for i, c in enumerate(nbhs):
            if c is nb:
                return i
        nbhs.append(nb)
        return len(nbhs) - 1

    data = {"n_calls": np.int32(len(calls)), "sample": np.int32(NET_SAMPLE)}
    for i, rec in enumerate(calls):
        mod, kw = rec["mod"], rec["kw"]
        ci, co, ni = cloud_id(kw["p_pc_in"]), cloud_id(kw["p_pc_out"]), nbh_id(kw["p_neighborhood"])
        w = mod.conv_weights_
        p = f"c{i:02d}/"
        data[p + "meta"] = np.array([convs.index(mod), ci, co, ni, w.shape[0], w.shape[2]], dtype=np.int32)
        data[p + "rho"], data[p + "nu"] = mod.norm_neigh_dist_.numpy(), mod.norm_num_neighs_.numpy()
        data[p + "param_sums"] = np.array([float(t.detach().double().sum()) for t in (mod.proj_axes_, mod.proj_biases_, w)] +
                                          [float(w.detach().double().abs().sum())])
        data[p + "x"] = kw["p_in_features"].detach().numpy()
        data[p + "grad_out"] = rec["grad_out"].numpy()
        for key, t in (("out", rec["out"].detach()), ("dx", rec["dx"])):
            if t is None:
                continue
            if net == "faust" or t.numel() <= 2 * OUT_SAMPLE:
                data[p + key] = t.numpy()
            else:  # seeded positions + the norm, like the large weight gradients
                pos = torch.randperm(t.numel(), generator=torch.Generator().manual_seed(9500 + 2 * i + (key == "dx")))[:OUT_SAMPLE]
                data[p + key + "_pos"], data[p + key + "_at"] = pos.numpy().astype(np.int32), t.reshape(-1)[pos].numpy()
                data[p + key + "_norm"], data[p + key + "_shape"] = np.float64(t.double().norm()), np.array(t.shape, dtype=np.int32)
        data[p + "dA"], data[p + "dbeta"] = mod.proj_axes_.grad.numpy(), mod.proj_biases_.grad.numpy()
        dw = w.grad
        if dw.numel() <= 4096:
            data[p + "dW"] = dw.numpy()
        else:
            pos = torch.randperm(dw.numel(), generator=torch.Generator().manual_seed(9000 + i))[:NET_SAMPLE]
            data[p + "dW_pos"], data[p + "dW_at"] = pos.numpy().astype(np.int32), dw.reshape(-1)[pos].numpy()
            data[p + "dW_norm"] = np.float64(dw.double().norm())
    for i, c in enumerate(clouds):
        data[f"cloud{i}/pts"], data[f"cloud{i}/batch"] = c.pts_.numpy(), c.batch_ids_.numpy().astype(np.int32)
        data[f"cloud{i}/frames"] = c.local_frames_.numpy()
    for i, nb in enumerate(nbhs):
        data[f"nbh{i}/neighbors"] = nb.neighbors_.numpy().astype(np.int32)
        data[f"nbh{i}/ends"] = nb.start_ids_.numpy().astype(np.int32)
        data[f"nbh{i}/radius"] = np.float64(nb.radius_)
    data["n_clouds"], data["n_nbhs"] = np.int32(len(clouds)), np.int32(len(nbhs))
    return data


PNE_CASES = [
    # name,              seed, n_in, n_out, c_in, c_out, k, batches
    ("pne_n300_c32",       9, 300, None, 32, 32, 16, 1),
    ("pne_down_n400_n150", 10, 400, 150, 3, 64, 20, 2),
]

CASES = [
    # name,             seed, n_in, n_out, F, c_in, c_out, k, batches
    ("cfg1_n1024_f1_c32", 0, 1024, None, 1, 32, 32, 16, 1),   # BASELINE.json configs[0]
    ("n256_f2_c64",       1, 256, None, 2, 64, 64, 16, 1),
    ("n256_f4_c32",       2, 256, None, 4, 32, 32, 12, 1),
    ("n256_f2_c1_c32",    3, 256, None, 2, 1, 32, 16, 1),     # DFaust input layer (C_in = 1)
    ("n256_f1_c3_c64",    4, 256, None, 1, 3, 64, 16, 1),     # ScanNet input layer (C_in = 3)
    ("down_n512_n128_f2", 5, 512, 128, 2, 32, 64, 24, 2),     # two clouds, two batches, ragged
    ("sparse_n200_f2",    6, 200, None, 2, 32, 32, 1.5, 1),   # many 1-neighbour (self only) rows
]


def main():
    pclib = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only in ("", "hierarchy"):
        path = os.path.join(OUT, "hierarchy.npz")
        np.savez_compressed(path, **hierarchy_case(pclib, 11))
        print(f"{path}: size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "hierarchy":
        return
    if only in ("", "grid_rnd"):
        path = os.path.join(OUT, "grid_rnd.npz")
        np.savez_compressed(path, **grid_rnd_case(pclib, 17))
        print(f"{path}: size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "grid_rnd":
        return
    if only in ("", "rel_rot"):
        for rel_rot, dims in (("matrix", 12), ("quaternion", 7)):
            path = os.path.join(OUT, f"rel_rot_{rel_rot}.npz")
            np.savez_compressed(path, **rel_rot_case(pclib, 19, rel_rot, dims))
            print(f"{path}: size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "rel_rot":
        return
    if only in ("", "keys"):
        import json
        path = os.path.join(OUT, "state_dict_keys.json")
        with open(path, "w") as fh:
            json.dump(state_dict_keys(pclib), fh, indent=1, sort_keys=True)
        print(f"{path}: written")
    if only == "keys":
        return
    if only in ("", "network"):
        path = os.path.join(OUT, "network_faust_calls.npz")
        data = network_case(pclib, 23)
        np.savez_compressed(path, **data)
        print("level points:", [int(data[f"cloud{i}/pts"].shape[0]) for i in range(int(data["n_clouds"]))])
        print(f"{path}: calls={int(data['n_calls'])} clouds={int(data['n_clouds'])} nbhs={int(data['n_nbhs'])} "
              f"size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "network":
        return
    if only in ("", "scannet"):
        path = os.path.join(OUT, "network_scannet_calls.npz")
        data = network_case(pclib, 29, "scannet")
        np.savez_compressed(path, **data)
        print("level points:", [int(data[f"cloud{i}/pts"].shape[0]) for i in range(int(data["n_clouds"]))])
        print(f"{path}: calls={int(data['n_calls'])} clouds={int(data['n_clouds'])} nbhs={int(data['n_nbhs'])} "
              f"size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "scannet":
        return
    if only in ("", "block"):
        path = os.path.join(OUT, "resnetformer_block.npz")
        np.savez_compressed(path, **block_case(pclib, 13))
        print(f"{path}: size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "block":
        return
    for name, seed, n_in, n_out, c_in, c_out, k, b in PNE_CASES:
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **pne_case(pclib, seed, n_in, n_out, c_in, c_out, k, b))
        print(f"{path}: size={os.path.getsize(path) / 1e6:.2f} MB")
    if only == "pne":
        return
    for name, seed, n_in, n_out, f, c_in, c_out, k, b in CASES:
        data = layer_case(pclib, seed, n_in, n_out, f, c_in, c_out, k, batches=b)
        path = os.path.join(OUT, f"layer_{name}.npz")
        np.savez_compressed(path, **data)
        print(f"{path}: E={data['neighbors'].shape[0]} rows_out={data['out'].shape[0]} "
              f"size={os.path.getsize(path) / 1e6:.2f} MB")
    np.savez_compressed(os.path.join(OUT, "rotation_fns.npz"), **rotation_case(pclib, 7))
    np.savez_compressed(os.path.join(OUT, "pca_frames.npz"), **pca_case(pclib, 8))
    print("done")


if __name__ == "__main__":
    main()
