"""Synthetic workloads of the benchmark and the size tests: clouds of stated
(N points, k neighbours, F frames, C channels) shaped like the batches the reference's task scripts feed
the layer.  Shared by ``bench.py``, ``tools/`` and ``tests/``; nothing here depends on the checker.

  headline        one cloud of 65 536 points, F = 2 random frames, C = 64, k ~ 32  -- BASELINE.json's metric
  scannet150k_f1  one ScanNet-like scene of 150 000 points, F = 1 frame about the fixed up axis, five levels of the widths the
                  ScanNet network runs, 64/128/192/256/320 (round 6; four levels of 64 before)
                  (tasks/SemSeg/confs/scannet/scannet20_rot_pca_SO2.yaml:5,26,40, seg_models.py:49-50; one scene per GPU = config 5)
  dfaust_f2       32 bodies x 2 200 points (4096 sampled, 0.04 grid), F = 2 PCA frames from 16-NN, level widths
                  32/64/128/256 (tasks/SemSeg/confs/dfaust/dfaust_I_rot_pca_2F.yaml:4,17,37-38; seg_models.py:26-27)
  dfaust_f4       16 bodies x 6 900 points, F = 4 PCA frames (the four sign flips of the eigenbasis)

Every workload is a stack: one same-level ``PNEConvLayerRotEquiv`` per hierarchy level, levels by grid
sub-sampling with cell doubling, radius = 2 x the cell that produced the level (seg_models.py:29-33).
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch

WORKLOADS: Dict[str, dict] = {
    "headline": dict(points=65536, clouds=1, frames=2, widths=[64, 64, 64, 64], degree=32, fixed_axis=False, pca=False,
                     note="4-level PNEConvLayerRotEquiv stack, N=65536, k=32, F=2, C=64"),
    "scannet150k_f1": dict(points=150000, clouds=1, frames=1, widths=[64, 128, 192, 256, 320], degree=32, fixed_axis=2, pca=False,
                           note="5-level stack on one ScanNet-like scene of 150k points, F=1 about the fixed up axis, C=64/128/192/256/320"),
    "dfaust_f2": dict(points=2200, clouds=32, frames=2, widths=[32, 64, 128, 256], degree=24, fixed_axis=False, pca=True,
                      note="4-level stack on a DFaust-like batch: 32 bodies x 2200 points, F=2 PCA frames, C=32/64/128/256"),
    "dfaust_f4": dict(points=6900, clouds=16, frames=4, widths=[32, 64, 128, 256], degree=24, fixed_axis=False, pca=True,
                      note="4-level stack on 16 bodies x 6900 points, F=4 PCA frames, C=32/64/128/256"),
}
NUM_BASIS = 32


def radius_for_degree(n: int, k: float) -> float:
    """r = (3k / (4 pi N))^(1/3): expected interior degree k for N uniform points in [0,1)^3."""
    return (3.0 * k / (4.0 * math.pi * n)) ** (1.0 / 3.0)


def frames_config(spec: dict) -> dict:
    cfg = {"pca": bool(spec["pca"]), "n_frames": spec["frames"], "fixed_axis": spec["fixed_axis"]}
    if spec["pca"]:
        cfg.update(neigh_method="knn", neigh_kwargs={"neigh_k": 16})
    return cfg


def morton_order(pts: torch.Tensor, bid: torch.Tensor, cell: float) -> torch.Tensor:
    """Permutation that sorts the points of every batch element along the Z-order curve of a grid of ``cell``-sized
    cells (what a loader does once per scene so that neighbouring points are neighbouring rows)."""
    q = torch.clamp((pts / cell).long(), 0, (1 << 20) - 1)

    def spread(v):      # 20 bits -> every third bit
        out = torch.zeros_like(v)
        for b in range(20):
            out |= ((v >> b) & 1) << (3 * b)
        return out

    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = torch.argsort(key, stable=True)
    return order[torch.argsort(bid[order].long(), stable=True)]


def build_cloud(spec: dict, device, seed: int, order: str = "random"):
    """Points ~ U[0,1)^3 per batch element, batch ids ascending, frames per ``spec`` (seeded).  ``order`` = "random"
    (rows in the order they were drawn) or "morton" (rows sorted along a Z-order curve per batch element)."""
    from . import pc as _pc

    torch.manual_seed(seed)
    n = spec["points"] * spec["clouds"]
    pts = torch.rand(n, 3, device=device)
    bid = torch.arange(spec["clouds"], device=device, dtype=torch.int32).repeat_interleave(spec["points"])
    if order == "morton":
        pts = pts[morton_order(pts, bid, radius_for_degree(spec["points"], spec["degree"]))]
    return _pc.PointcloudRotEquiv(pts, bid, frames_config(spec))


def build_stack(spec: dict, device, seed: int, n_levels: int = 0, order: str = "random") -> List[dict]:
    """The stack of one rank: per level the cloud, its ball-query neighbourhood, a conv with converged EMA
    buffers (rho = 1/r, nu = M/E), input features and an output gradient.  n_levels = 0: one level per entry of the
    workload's widths."""
    from . import layers, pc as _pc

    n_levels = n_levels or len(spec["widths"])

    r0 = radius_for_degree(spec["points"], spec["degree"])
    pc0 = build_cloud(spec, device, seed, order)
    cells = [r0 * 2 ** i for i in range(n_levels - 1)]
    hier = _pc.PointHierarchyRotEquiv(pc0, n_levels - 1, "grid_avg", grid_radii=cells)
    radii = [r0 * 2 ** i for i in range(n_levels)]
    factory = layers.PNEConvLayerRotEquivFactory(9, NUM_BASIS, "mlp_gelu")
    f = spec["frames"]
    levels = []
    for lvl, (pc, r) in enumerate(zip(hier.pcs_, radii)):
        ch = spec["widths"][lvl]
        nbh = hier.create_neighborhood(lvl, lvl, "ball_query", bq_radius=r)
        conv = factory.create_conv_layer(ch, ch).to(device)
        conv.norm_neigh_dist_.fill_(1.0 / r)
        conv.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / max(nbh.num_edges(), 1))
        n = pc.pts_.shape[0]
        x = torch.randn(n * f, ch, device=device, requires_grad=True)
        g = torch.randn(n * f, ch, device=device)
        levels.append(dict(pc=pc, nbh=nbh, conv=conv, x=x, g=g, n=n, e=nbh.num_edges(), r=r, c=ch, f=f))
    return levels


def build_level_pair(spec: dict, device, seed: int, order: str = "random"):
    """Levels 0 and 1 of the workload's hierarchy with their radii: the clouds of the encoder's first down-convolution
    (level 0 -> 1, neighbourhood radius of the SOURCE level, models/Encoder.py:137-147,167-171) and of the decoder's last
    up-convolution (level 1 -> 0, radius of the COARSER level, models/Decoder.py:66-74,83-88)."""
    from . import pc as _pc

    r0 = radius_for_degree(spec["points"], spec["degree"])
    pc0 = build_cloud(spec, device, seed, order)
    hier = _pc.PointHierarchyRotEquiv(pc0, 1, "grid_avg", grid_radii=[r0])
    return hier.pcs_[0], hier.pcs_[1], r0, 2.0 * r0


def build_down_up(spec: dict, device, seed: int, order: str = "random") -> List[dict]:
    """The two level-to-level convolutions of ``build_level_pair`` as bench records (like ``build_stack``'s, with both
    clouds): a ball-query neighbourhood between the levels, a conv with converged EMA buffers, features and an output
    gradient.  Channels follow the workload's level widths (down: widths[0] -> widths[1], up: the reverse)."""
    from . import layers, pc as _pc

    pc0, pc1, r0, r1 = build_level_pair(spec, device, seed, order)
    factory = layers.PNEConvLayerRotEquivFactory(9, NUM_BASIS, "mlp_gelu")
    f = spec["frames"]
    w0, w1 = spec["widths"][0], spec["widths"][1]
    recs = []
    for name, pc_in, pc_out, r, c_in, c_out in (("down", pc0, pc1, r0, w0, w1), ("up", pc1, pc0, r1, w1, w0)):
        nbh = _pc.BQNeighborhood(pc_in, pc_out, r)
        conv = factory.create_conv_layer(c_in, c_out).to(device)
        conv.norm_neigh_dist_.fill_(1.0 / r)
        conv.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / max(nbh.num_edges(), 1))
        n_in, n_out = pc_in.pts_.shape[0], pc_out.pts_.shape[0]
        x = torch.randn(n_in * f, c_in, device=device, requires_grad=True)
        g = torch.randn(n_out * f, c_out, device=device)
        recs.append(dict(name=name, pc_in=pc_in, pc_out=pc_out, nbh=nbh, conv=conv, x=x, g=g, n_in=n_in, n_out=n_out,
                         e=nbh.num_edges(), r=r, c_in=c_in, c_out=c_out, f=f))
    return recs


def layer_bytes_two_clouds(n_in: int, n_out: int, e: int, f_in: int, f_out: int, c_in: int, c_out: int, kb: int = NUM_BASIS) -> int:
    """SURVEY.md section 8d's algorithmic bytes of one layer forward + backward for N_in != N_out, as written there:
    B_f = 8E + 4N_out + 12(N_in + N_out) + 36(F_in N_in + F_out N_out) + 4 E F_in C_in + 4 M' C_out + 4(10K + C_in K C_out),
    B_b = the same geometry + 4 E F_in C_in (re-gather of f) + 4 E F_in C_in (dX by the source-sorted segmented reduce)
          + 4 M' C_out (g) + 4 N_in F_in C_in (dX) + 8(10K + C_in K C_out).  Equals ``layer_bytes`` for one cloud."""
    geom = 8 * e + 4 * n_out + 12 * (n_in + n_out) + 36 * (f_in * n_in + f_out * n_out)
    params = 4 * (10 * kb + c_in * kb * c_out)
    gather = 4 * e * f_in * c_in
    b_f = geom + gather + 4 * n_out * f_out * c_out + params
    b_b = geom + 2 * gather + 4 * n_out * f_out * c_out + 4 * n_in * f_in * c_in + 2 * params
    return b_f + b_b


# ---- algorithmic work of one layer ---------------------------------------------------
def layer_flops(n: int, e: int, f: int, c: int, kb: int = NUM_BASIS) -> Dict[str, int]:
    """Algorithmic FLOPs per stage (MLP counted with its bias row)."""
    ep, rows = e * f * f, n * f
    dense = rows * 2 * c * kb * c
    edge = ep * (2 * 10 * kb + 2 * c * kb)
    pg = ep * (2 * 10 * kb + 2 * c * kb + 2 * 10 * kb)
    return {"edge_t_fwd": edge, "gemm_out": dense, "gemm_gradT": dense, "gemm_gradW": dense, "gemm_gradX": dense,
            "edge_t_transposed": edge, "edge_param_grad": pg}


def stage_owned_bytes(n: int, e: int, f: int, c: int, kb: int = NUM_BASIS) -> Dict[str, int]:
    """SURVEY.md section 8d's algorithmic bytes of one layer fwd+bwd, split over the launches that own them
    (uncached-gather model: every point-edge touches its neighbour's F*C block once per pass; no T, no
    E'-sized or row-sized intermediates).  The values sum to ``layer_bytes``."""
    rows = n * f
    geom = 8 * e + 4 * n + 12 * 2 * n + 36 * 2 * rows
    params = 4 * (10 * kb + c * kb * c)
    gather = 4 * e * f * c
    act = 4 * rows * c
    return {"edge_t_fwd": geom + gather,            # B_f: edges, ends, points, frames, gathered f
            "gemm_out": act + params,               # out write, parameters
            "edge_t_transposed": gather,            # dX via the source-sorted segmented reduce: gathered g
            "gemm_gradX": act,                      # dX write
            "edge_param_grad": geom + gather,       # B_b: geometry again, re-gather of f
            "gemm_gradT": act + params,             # g read, parameters
            "gemm_gradW": params}                   # parameter gradients written


def layer_bytes(n: int, e: int, f: int, c: int, kb: int = NUM_BASIS) -> int:
    return sum(stage_owned_bytes(n, e, f, c, kb).values())


def stage_moved_bytes(n: int, e: int, f: int, c: int, bytes_per_el=(3, 3, 4), kb: int = NUM_BASIS) -> Dict[str, int]:
    """What each launch of the unfused pipeline has to move at least: its owned bytes plus the row-sized
    intermediates it writes or reads.  ``bytes_per_el`` = bytes per element of (T, U, grad_T) as the library reports
    them for the shape (``se3conv_intermediate_bytes_per_element``: 3-byte rows, 4-byte words, 0 = never written)."""
    rows = n * f
    t_bytes, u_bytes, g_bytes = (int(b * rows * c * kb) for b in bytes_per_el)  # (2.25 for the T16 block format)
    own = stage_owned_bytes(n, e, f, c, kb)
    act = 4 * rows * c
    return {"edge_t_fwd": own["edge_t_fwd"] + t_bytes, "gemm_out": own["gemm_out"] + t_bytes,
            "edge_t_transposed": own["edge_t_transposed"] + u_bytes, "gemm_gradX": own["gemm_gradX"] + u_bytes,
            "edge_param_grad": own["edge_param_grad"] + g_bytes, "gemm_gradT": own["gemm_gradT"] + g_bytes,
            "gemm_gradW": own["gemm_gradW"] + t_bytes + act}


def stage_gather_bytes(n: int, e: int, f: int, c: int) -> Dict[str, int]:
    """The part of ``stage_owned_bytes`` that is the uncached-gather model's neighbour rows (every point-edge fetching its
    neighbour's F*C block).  The gathered table itself is rows * C * 4 bytes (33.5 MB at the headline shape): it lives
    in the 256 MiB memory-side cache for the whole launch, so these bytes are served on chip and are NOT HBM traffic."""
    gather = 4 * e * f * c
    return {"edge_t_fwd": gather, "edge_t_transposed": gather, "edge_param_grad": gather}


def stage_hbm_bytes(n: int, e: int, f: int, c: int, bytes_per_el=(3, 3, 4), kb: int = NUM_BASIS) -> Dict[str, int]:
    """What each launch must move through HBM at least: ``stage_moved_bytes`` with the cache-served gathers replaced by one
    read of the gathered table (row-sized intermediates + compulsory bytes)."""
    moved = stage_moved_bytes(n, e, f, c, bytes_per_el, kb)
    table = 4 * n * f * c
    return {t: v - stage_gather_bytes(n, e, f, c).get(t, 0) + (table if t in ("edge_t_fwd", "edge_t_transposed", "edge_param_grad") else 0)
            for t, v in moved.items()}


# ---- BASELINE configs 2 / 3 at their own shapes: the convolution calls of the reference's FAUST / ScanNet networks -----------
def faust_network_calls(fixture_path: str) -> List[dict]:
    """The call list of the reference's FPNSegUNetMLPGeluRotEqFAUST as recorded from the reference itself
    (tests/golden/network_faust_calls.npz, tools/gen_golden.py `network_case`): per call the hierarchy level of the input
    and of the output cloud (5 = the task script's randomly sub-sampled output cloud), the neighbourhood radius and the
    channel counts -- patch encoder (0 -> 1, 1 -> 1), two blocks per level 1..4 with the down-convolutions between them,
    the decoder's up-convolutions, the three FPN laterals onto level 1, the patch decoder (1 -> 0) and the head (0 -> out)."""
    import numpy as np

    calls = []
    with np.load(fixture_path) as z:
        for i in range(int(z["n_calls"])):
            _, ci, co, ni, c_in, c_out = (int(v) for v in z[f"c{i:02d}/meta"])
            calls.append(dict(level_in=ci, level_out=co, radius=float(z[f"nbh{ni}/radius"]), c_in=c_in, c_out=c_out))
    return calls


def faust_raw_batch(device, bodies: int = 32, sampled: int = 4096, seed: int = 0):
    """A DFaust-sized batch as the loader hands it over: ``sampled`` points per body on thin shells of a torso's
    proportions, bodies 3 m apart, one batch id per body."""
    torch.manual_seed(seed)
    u = torch.randn(bodies * sampled, 3, device=device)
    u = u / u.norm(dim=1, keepdim=True)
    scale = 1.0 + 0.15 * torch.rand(bodies, 1, device=device).repeat_interleave(sampled, 0)
    pts = u * torch.tensor([0.27, 0.9, 0.17], device=device) * scale + 0.01 * torch.randn(bodies * sampled, 3, device=device)
    pts = pts + torch.arange(bodies, device=device, dtype=torch.float32).repeat_interleave(sampled)[:, None] * torch.tensor([3.0, 0.0, 0.0], device=device)
    bid = torch.arange(bodies, device=device, dtype=torch.int32).repeat_interleave(sampled)
    return pts, bid


def faust_clouds(pts: torch.Tensor, bid: torch.Tensor) -> list:
    """create_hierarchy of the task script (tasks/SemSeg/train_dfaust_rot.py:108-158, confs/dfaust/dfaust_I_rot_pca_2F.yaml):
    init sub-sample 0.04, PCA frames from 16-NN (F = 2) on every level, grid sub-samples 0.05 .. 0.4, and the randomly
    sub-sampled output cloud -> [level 0 .. level 4, output cloud]."""
    from . import pc as _pc

    cfg = {"pca": True, "n_frames": 2, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
    raw = _pc.Pointcloud(pts, bid)
    samp = _pc.GridSubSample(raw, 0.04)
    pc0 = _pc.PointcloudRotEquiv(samp.__subsample_tensor__(raw.pts_, "avg"), samp.__subsample_tensor__(raw.batch_ids_, "max"), cfg)
    hier = _pc.PointHierarchyRotEquiv(pc0, 4, "grid_avg", grid_radii=[0.05, 0.1, 0.2, 0.4])
    samp_out = _pc.GridSubSample(raw, 0.04, p_rnd_sample=True)
    out_pc = _pc.PointcloudRotEquiv(samp_out.__subsample_tensor__(raw.pts_, "avg"),
                                    samp_out.__subsample_tensor__(raw.batch_ids_, "max"), cfg)
    return list(hier.pcs_) + [out_pc]


def faust_neighbourhoods(clouds: list, calls: List[dict], capacities: dict = None) -> dict:
    """The neighbourhoods the network's calls use, one per (input level, output level, radius) as the reference's hierarchy
    memoises them (PointHierarchy.py:60-79), with the source-major list backward reads where the clouds differ.
    ``capacities`` (same keys -> rows): capacity-bounded builds without a host synchronisation."""
    from . import pc as _pc

    nbhs = {}
    for c in calls:
        key = (c["level_in"], c["level_out"], c["radius"])
        if key in nbhs:
            continue
        cap = None if capacities is None else capacities[key]
        nb = _pc.BQNeighborhood(clouds[c["level_in"]], clouds[c["level_out"]], c["radius"], p_capacity=cap)
        if c["level_in"] != c["level_out"] and cap is not None:
            nb.source_major()
        nbhs[key] = nb
    return nbhs


def build_faust_network_convs(device, fixture_path: str, bodies: int = 32, sampled: int = 4096, seed: int = 0) -> List[dict]:
    """The hierarchy the task script builds for a DFaust batch (tasks/SemSeg/train_dfaust_rot.py:108-158 with
    confs/dfaust/dfaust_I_rot_pca_2F.yaml: 4096 sampled points per body, init / output sub-sample 0.04, grid sub-samples
    0.05 .. 0.4, PCA frames from 16-NN, F = 2) on synthetic bodies -- thin shells of a torso's proportions, ~2 k points per
    body at level 0 -- and one bench record per convolution call of the network (``faust_network_calls``): clouds,
    ball-query neighbourhood with the call's radius, a layer with converged EMA buffers, features and an output gradient."""
    from . import layers, pc as _pc

    pts, bid = faust_raw_batch(device, bodies, sampled, seed)
    clouds = faust_clouds(pts, bid)
    factory = layers.PNEConvLayerRotEquivFactory(9, NUM_BASIS, "mlp_gelu")
    calls = faust_network_calls(fixture_path)
    nbhs = faust_neighbourhoods(clouds, calls)  # the network builds a neighbourhood once and shares it between the convolutions that use it
    recs = []
    for i, c in enumerate(calls):
        pc_in, pc_out = clouds[c["level_in"]], clouds[c["level_out"]]
        nbh = nbhs[(c["level_in"], c["level_out"], c["radius"])]
        conv = factory.create_conv_layer(c["c_in"], c["c_out"]).to(device)
        conv.norm_neigh_dist_.fill_(1.0 / c["radius"])
        conv.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / max(nbh.num_edges(), 1))
        n_in, n_out = pc_in.pts_.shape[0], pc_out.pts_.shape[0]
        # the network's first convolution reads the input features (no gradient); every other one is fed by a layer
        x = torch.randn(n_in * 2, c["c_in"], device=device, requires_grad=i > 0)
        g = torch.randn(n_out * 2, c["c_out"], device=device)
        recs.append(dict(name=f"call{i:02d}", pc_in=pc_in, pc_out=pc_out, nbh=nbh, conv=conv, x=x, g=g, n_in=n_in, n_out=n_out,
                         e=nbh.num_edges(), r=c["radius"], c_in=c["c_in"], c_out=c["c_out"], f=2, level_in=c["level_in"],
                         level_out=c["level_out"]))
    return recs


def build_scannet_network_convs(device, fixture_path: str, scenes: int = 6, raw_points: int = 120000, seed: int = 0) -> List[dict]:
    """BASELINE config 3 at the network's own shapes: the hierarchy the task script builds for a ScanNet batch
    (tasks/SemSeg/train_scannet_rot.py:142-186 with confs/scannet/scannet20_rot_pca_SO2.yaml: scenes of up to 120 000 points,
    init / output sub-sample 0.1, grid sub-samples 0.2 .. 1.6, PCA frames from 16-NN about the fixed axis 2, F = 1) on
    synthetic rooms -- floor, ceiling, four walls and a few boxes of furniture, points on the surfaces with scanner noise --
    and one bench record per convolution call of FPNSegUNetMLPGeluRotEqScanNet (call list recorded from the reference,
    tests/golden/network_scannet_calls.npz: 32 calls, widths 64 .. 320, FPN laterals of width 128 onto level 0)."""
    from . import layers, pc as _pc

    gen = torch.Generator(device="cpu").manual_seed(seed)
    parts, bids = [], []
    for sc in range(scenes):
        w, d, hgt = (float(v) for v in (torch.rand(3, generator=gen) * torch.tensor([4.0, 3.0, 0.6]) + torch.tensor([5.0, 4.0, 2.4])))
        surf = [([0, 0, 0], [w, 0, 0], [0, d, 0]), ([0, 0, hgt], [w, 0, 0], [0, d, 0]),     # floor, ceiling
                ([0, 0, 0], [w, 0, 0], [0, 0, hgt]), ([0, d, 0], [w, 0, 0], [0, 0, hgt]),     # walls
                ([0, 0, 0], [0, d, 0], [0, 0, hgt]), ([w, 0, 0], [0, d, 0], [0, 0, hgt])]
        for _ in range(6):  # furniture: top and two sides of a box
            bx, by = (float(v) for v in torch.rand(2, generator=gen) * torch.tensor([w - 1.6, d - 1.2]))
            bw, bd, bh = (float(v) for v in (torch.rand(3, generator=gen) * torch.tensor([1.0, 0.7, 0.6]) + torch.tensor([0.5, 0.4, 0.4])))
            surf += [([bx, by, bh], [bw, 0, 0], [0, bd, 0]), ([bx, by, 0], [bw, 0, 0], [0, 0, bh]), ([bx, by, 0], [0, bd, 0], [0, 0, bh])]
        area = torch.tensor([torch.linalg.cross(torch.tensor(u, dtype=torch.float32), torch.tensor(v, dtype=torch.float32)).norm() for _, u, v in surf])
        counts = torch.round(area / area.sum() * raw_points).long()
        for (o, u, v), n in zip(surf, counts.tolist()):
            ab = torch.rand(n, 2, generator=gen)
            parts.append(torch.tensor(o, dtype=torch.float32) + ab[:, :1] * torch.tensor(u, dtype=torch.float32) + ab[:, 1:] * torch.tensor(v, dtype=torch.float32)
                         + torch.tensor([12.0 * sc, 0.0, 0.0]))
            bids.append(torch.full((n,), sc, dtype=torch.int32))
    pts = torch.cat(parts)
    pts = (pts + 0.004 * torch.randn(pts.shape, generator=gen)).to(device)
    bid = torch.cat(bids).to(device)
    cfg = {"pca": True, "n_frames": 1, "fixed_axis": 2, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
    raw = _pc.Pointcloud(pts, bid)
    samp = _pc.GridSubSample(raw, 0.1)
    pc0 = _pc.PointcloudRotEquiv(samp.__subsample_tensor__(raw.pts_, "avg"), samp.__subsample_tensor__(raw.batch_ids_, "max"), cfg)
    hier = _pc.PointHierarchyRotEquiv(pc0, 4, "grid_avg", grid_radii=[0.2, 0.4, 0.8, 1.6])
    samp_out = _pc.GridSubSample(raw, 0.1, p_rnd_sample=True)
    out_pc = _pc.PointcloudRotEquiv(samp_out.__subsample_tensor__(raw.pts_, "avg"),
                                    samp_out.__subsample_tensor__(raw.batch_ids_, "max"), cfg)
    clouds = list(hier.pcs_) + [out_pc]
    factory = layers.PNEConvLayerRotEquivFactory(9, NUM_BASIS, "mlp_gelu")
    recs, nbhs = [], {}
    for i, c in enumerate(faust_network_calls(fixture_path)):  # (the call-list reader is the same for both fixtures)
        pc_in, pc_out = clouds[c["level_in"]], clouds[c["level_out"]]
        key = (c["level_in"], c["level_out"], c["radius"])
        if key not in nbhs:
            nbhs[key] = _pc.BQNeighborhood(pc_in, pc_out, c["radius"])
        nbh = nbhs[key]
        conv = factory.create_conv_layer(c["c_in"], c["c_out"]).to(device)
        conv.norm_neigh_dist_.fill_(1.0 / c["radius"])
        conv.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / max(nbh.num_edges(), 1))
        n_in, n_out = pc_in.pts_.shape[0], pc_out.pts_.shape[0]
        x = torch.randn(n_in, c["c_in"], device=device, requires_grad=True)  # the first convolution is fed by the input embedding
        g = torch.randn(n_out, c["c_out"], device=device)
        recs.append(dict(name=f"call{i:02d}", pc_in=pc_in, pc_out=pc_out, nbh=nbh, conv=conv, x=x, g=g, n_in=n_in, n_out=n_out,
                         e=nbh.num_edges(), r=c["radius"], c_in=c["c_in"], c_out=c["c_out"], f=1, level_in=c["level_in"],
                         level_out=c["level_out"]))
    return recs
