"""GPU: the convolutions BETWEEN two hierarchy levels at BASELINE.json's full sizes -- the encoder's down-convolution
(level l -> l + 1, models/Encoder.py:167-171: radius of the source level) and the decoder's up-convolution (level l ->
l - 1, models/Decoder.py:57-100: radius of the coarser level) -- against the oracle by restriction, like
test_gpu_fullsize_backward.py does for the same-level layers:

  * the output rows of a set S of samples, and d[A; beta] / dW with grad_out zeroed outside S, only involve the edges
    INTO S: the oracle runs on (samples S, sources = every point with an edge into S);
  * dX of a set P of source points only involves the edges that LEAVE P: the oracle runs on (sources P, samples =
    every point with an edge from P).

N_in != N_out, F on both sides, a non-symmetric edge relation: backward reads the source-major copy of the edge list
(se3_csr_transpose), which the same-level tests never build.  `headline` (65 536 -> ~9 200 points, 64 channels) and
`dfaust_f2` (32 bodies, PCA frames, 32 <-> 64 channels); both arithmetic modes."""
import pytest
import torch

from conftest import canon_edges, rel_err
from oracle import se3conv_oracle as O
from se3conv3d_amd import workloads as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOLS = {"fp32": 2e-5, "bf16x3": 5e-5, "bf16x3_t16": 5e-5}  # the T16 mode is held to the same bound as bf16x3


@pytest.fixture(scope="module", params=["bf16x3", "fp32", "bf16x3_t16"])
def amd(built_library, request):
    import se3conv3d_amd

    se3conv3d_amd.set_precision(request.param)
    yield se3conv3d_amd
    se3conv3d_amd.set_precision("bf16x3")


def rows_of(points: torch.Tensor, f: int) -> torch.Tensor:
    return (points[:, None] * f + torch.arange(f, device=points.device)[None, :]).reshape(-1)


def oracle_sub(pc_in, pc_out, conv, nb_sub, samples, sources, x_rows, g_rows):
    """Oracle forward + backward on a sub-graph; `nb_sub` [E,2] in ORIGINAL ids sorted by sample."""
    s_map = torch.full((pc_out.pts_.shape[0],), -1, dtype=torch.int64)
    p_map = torch.full((pc_in.pts_.shape[0],), -1, dtype=torch.int64)
    s_map[samples] = torch.arange(samples.shape[0])
    p_map[sources] = torch.arange(sources.shape[0])
    nb = torch.stack((s_map[nb_sub[:, 0]], p_map[nb_sub[:, 1]]), 1)
    assert int(nb.min()) >= 0 and bool((nb[1:, 0] >= nb[:-1, 0]).all())
    cpu = lambda t: t.detach().cpu()
    return O.conv_forward_backward(cpu(pc_in.pts_)[sources], cpu(pc_out.pts_)[samples], cpu(pc_in.local_frames_)[sources],
                                   cpu(pc_out.local_frames_)[samples], nb, x_rows, cpu(conv.proj_axes_), cpu(conv.proj_biases_),
                                   cpu(conv.conv_weights_), cpu(conv.norm_neigh_dist_), cpu(conv.norm_num_neighs_), g_rows)


def gpu_step(conv, pc_in, pc_out, nbh, x, g):
    for p in conv.parameters():
        p.grad = None
    xg = x.clone().requires_grad_(True)
    out = conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=xg, p_neighborhood=nbh)
    out.backward(g)
    return out.detach(), xg.grad, conv.proj_axes_.grad.clone(), conv.proj_biases_.grad.clone(), conv.conv_weights_.grad.clone()


def check_two_cloud_layer(amd, pc_in, pc_out, r, c_in, c_out, seed):
    tol = TOLS[amd.get_precision()]
    torch.manual_seed(seed)
    f_in, f_out = pc_in.n_frames_, pc_out.n_frames_
    n_in, n_out = pc_in.pts_.shape[0], pc_out.pts_.shape[0]
    nbh = amd.pc.BQNeighborhood(pc_in, pc_out, r)
    assert not nbh.symmetric_
    nb = nbh.neighbors_.cpu()
    # the edge set itself, on the samples the brute-force oracle can afford
    some = torch.randperm(n_out)[:64].sort().values
    nb_o, ends_o = O.ball_query(pc_in.pts_.cpu(), pc_out.pts_.cpu()[some], pc_in.batch_ids_.cpu(), pc_out.batch_ids_.cpu()[some], r)
    keep = torch.zeros(n_out, dtype=torch.bool)
    keep[some] = True
    mine = nb[keep[nb[:, 0]]].clone()
    nb_o = torch.stack((some[nb_o[:, 0]], nb_o[:, 1]), 1)
    assert torch.equal(canon_edges(mine), canon_edges(nb_o))

    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c_in, c_out).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r)
    conv.norm_num_neighs_.fill_(n_out / nb.shape[0])
    with torch.no_grad():
        conv.proj_biases_.uniform_(-0.5, 0.5)
    x = torch.randn(n_in * f_in, c_in, device=DEV)
    g = torch.randn(n_out * f_out, c_out, device=DEV)
    out, dx, da, db, dw = gpu_step(conv, pc_in, pc_out, nbh, x, g)
    assert all(bool(torch.isfinite(t).all()) for t in (out, dx, da, db, dw))

    # the source-major edge list backward used: a permutation of the edges grouped by source
    geom = nbh._se3_geom[1] if hasattr(nbh, "_se3_geom") else None
    if geom is not None:
        ts, te = geom.transpose()
        deg_in = torch.bincount(nb[:, 1], minlength=n_in)
        assert torch.equal(te.cpu().long(), torch.cumsum(deg_in, 0))
        srcs = torch.repeat_interleave(torch.arange(n_in), deg_in)
        assert torch.equal(canon_edges(torch.stack((ts.cpu().long(), srcs), 1)), canon_edges(nb))

    # dX rows of source slices (first, interior, last)
    for p0 in (0, n_in // 2 + 5, n_in - 32):
        sources = torch.arange(p0, p0 + 32)
        m = (nb[:, 1] >= p0) & (nb[:, 1] < p0 + 32)
        nb_sub = nb[m]
        if nb_sub.shape[0] == 0:
            continue
        samples = torch.unique(nb_sub[:, 0])
        ref = oracle_sub(pc_in, pc_out, conv, nb_sub, samples, sources, x.cpu()[rows_of(sources, f_in)], g.cpu()[rows_of(samples, f_out)])
        got = dx[rows_of(sources.to(DEV), f_in)]
        assert rel_err(got, ref[1]) < tol, ("dX slice", p0, rel_err(got, ref[1]))

    # output rows + masked parameter gradients of sample sets (spread over the cloud; the last rows)
    for samples in (torch.randperm(n_out)[:64].sort().values, torch.arange(n_out - 24, n_out)):
        g_m = torch.zeros_like(g)
        rows = rows_of(samples.to(DEV), f_out)
        g_m[rows] = g[rows]
        out_m, _, da_m, db_m, dw_m = gpu_step(conv, pc_in, pc_out, nbh, x, g_m)
        keep = torch.zeros(n_out, dtype=torch.bool)
        keep[samples] = True
        nb_sub = nb[keep[nb[:, 0]]]
        sources = torch.unique(nb_sub[:, 1])
        ref = oracle_sub(pc_in, pc_out, conv, nb_sub, samples, sources, x.cpu()[rows_of(sources, f_in)], g.cpu()[rows_of(samples, f_out)])
        assert rel_err(out_m[rows], ref[0]) < tol, ("out rows", rel_err(out_m[rows], ref[0]))
        for name, u, v in (("dA", da_m, ref[2]), ("dbeta", db_m, ref[3]), ("dW", dw_m, ref[4])):
            assert rel_err(u, v) < tol, (name, rel_err(u, v))

    # adjoint identity at full size: <out, g> == <x, dX>.  Both sides are sums of millions of terms of either sign, so the
    # yardstick is the size of the terms (root sum of squares), not the size of the sum
    og, xd = out.double() * g.double(), x.double() * dx.double()
    lhs, rhs = float(og.sum()), float(xd.sum())
    scale = float(og.pow(2).sum().sqrt() + xd.pow(2).sum().sqrt())
    assert abs(lhs - rhs) <= 4 * tol * scale, (lhs, rhs, scale)


@pytest.mark.parametrize("workload", ["headline", "dfaust_f2"])
@pytest.mark.parametrize("direction", ["down", "up"])
def test_level_to_level_convolution_against_oracle(amd, workload, direction):
    spec = W.WORKLOADS[workload]
    pc0, pc1, r0, r1 = W.build_level_pair(spec, DEV, seed=0)
    c0, c1 = spec["widths"][0], spec["widths"][1]
    if direction == "down":
        check_two_cloud_layer(amd, pc0, pc1, r0, c0, c1, seed=11)
    else:
        check_two_cloud_layer(amd, pc1, pc0, r1, c1, c0, seed=12)


@pytest.mark.parametrize("f,c_in,c_out", [(2, 32, 64), (2, 64, 64), (4, 32, 32), (2, 128, 64)])
def test_edge_major_feature_gradient_of_a_down_convolution(amd, f, c_in, c_out):
    """A convolution with many more input than output rows takes the edge-major feature gradient (edge_dx.hip: D = phi gT^T per
    frame-edge, summed per source point) in the split-bf16 modes: dX against the oracle, with and without the transposition's
    edge ids (without them the kernel looks every edge up in the sample's neighbour list: same rows, same order, same bits),
    for one and two pairs of centre frames, 32-channel rows and 64-channel blocks, and with the parameter gradients off
    (grad_T is then computed for this path alone)."""
    from se3conv3d_amd import ops
    torch.manual_seed(f * 100 + c_in)
    n_in, n_out = 6000, 700
    cfg = {"pca": False, "n_frames": f, "fixed_axis": False}
    pc_in = amd.pc.PointcloudRotEquiv(torch.rand(n_in, 3, device=DEV), torch.zeros(n_in, dtype=torch.int32, device=DEV), cfg)
    pc_out = amd.pc.PointcloudRotEquiv(torch.rand(n_out, 3, device=DEV), torch.zeros(n_out, dtype=torch.int32, device=DEV), cfg)
    r = W.radius_for_degree(n_in, 18)
    nbh = amd.pc.BQNeighborhood(pc_in, pc_out, r)
    e = nbh.num_edges()
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c_in, c_out).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r), conv.norm_num_neighs_.fill_(n_out / e)
    x = torch.randn(n_in * f, c_in, device=DEV)
    g = torch.randn(n_out * f, c_out, device=DEV)
    out, dx, da, db, dw = gpu_step(conv, pc_in, pc_out, nbh, x, g)
    cpu = lambda t: t.detach().cpu()
    ref = O.conv_forward_backward(cpu(pc_in.pts_), cpu(pc_out.pts_), cpu(pc_in.local_frames_), cpu(pc_out.local_frames_),
                                  cpu(nbh.neighbors_), cpu(x), cpu(conv.proj_axes_), cpu(conv.proj_biases_), cpu(conv.conv_weights_),
                                  1.0 / r, n_out / e, cpu(g))
    tol = TOLS[ops._precision]
    for got, want, name in zip((out, dx, da, db, dw), ref, ("out", "dX", "dA", "dbeta", "dW")):
        assert rel_err(got, want) < tol, name
    # the same backward without the edge ids, and with the parameter gradients off
    geom = nbh._se3_geom[1]
    assert geom._edge_ids is not None, "the library's transposition records where every entry came from"
    ids, geom._edge_ids = geom._edge_ids, None
    try:
        _, dx_scan, *_ = gpu_step(conv, pc_in, pc_out, nbh, x, g)
    finally:
        geom._edge_ids = ids
    assert torch.equal(dx_scan, dx)
    for p in conv.parameters():
        p.requires_grad_(False)
    xg = x.clone().requires_grad_(True)
    conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=xg, p_neighborhood=nbh).backward(g)
    assert rel_err(xg.grad, ref[1]) < tol


@pytest.mark.parametrize("kind", ["same", "up", "down"])
def test_weight_gradient_from_u_and_from_t_agree(amd, kind):
    """Round 5: se3conv_bwd takes dW from U (the transposed pass's tensor: dW[i,k,o] = alpha sum_p f[p,i] U[p,o,k]) when the
    forward pass kept no T, and also with a T at hand when U is the smaller tensor (an up-convolution); from T otherwise.
    Raw calls with and without `t_save`: both results against the oracle, `se3conv_bwd_needs_t` consistent with what the module
    saves, and without the feature gradient (no U) the T path is what runs."""
    import ctypes as C

    from se3conv3d_amd import _lib, layers, ops
    if ops._precision == "fp32":
        pytest.skip("the exact-fp32 mode always keeps T")
    torch.manual_seed(40)
    f, c_in, c_out = 2, 64, 64
    n_a, n_b = 3000, 900
    cfg = {"pca": False, "n_frames": f, "fixed_axis": False}
    pc_a = amd.pc.PointcloudRotEquiv(torch.rand(n_a, 3, device=DEV), torch.zeros(n_a, dtype=torch.int32, device=DEV), cfg)
    pc_b = amd.pc.PointcloudRotEquiv(torch.rand(n_b, 3, device=DEV), torch.zeros(n_b, dtype=torch.int32, device=DEV), cfg)
    pc_in, pc_out = {"same": (pc_a, pc_a), "up": (pc_b, pc_a), "down": (pc_a, pc_b)}[kind]
    r = W.radius_for_degree(n_a if kind != "up" else n_b, 28)  # (denser than 20 edges per source: the same-level case keeps the U form)
    nbh = amd.pc.BQNeighborhood(pc_in, pc_out, r)
    n_in, n_out, e = pc_in.pts_.shape[0], pc_out.pts_.shape[0], nbh.num_edges()
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c_in, c_out).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r), conv.norm_num_neighs_.fill_(n_out / e)
    with torch.no_grad():
        conv.proj_biases_.uniform_(-0.5, 0.5)
    x = torch.randn(n_in * f, c_in, device=DEV)
    g = torch.randn(n_out * f, c_out, device=DEV)
    geom = layers._geometry_of(pc_in, pc_out, nbh)
    args = (geom, x, conv.proj_axes_.detach(), conv.proj_biases_.detach(), conv.conv_weights_.detach(), conv.norm_neigh_dist_,
            conv.norm_num_neighs_)
    out, t_save = ops.se3conv_forward(*args, save_t=True)
    out2, none = ops.se3conv_forward(*args, save_t=False)
    assert none is None and torch.equal(out, out2)
    with_t = ops.se3conv_backward(*args, t_save, g)
    without_t = ops.se3conv_backward(*args, None, g)
    params_only = ops.se3conv_backward(*args, t_save, g, want_feat=False)
    cpu = lambda t: t.detach().cpu()
    ref = O.conv_forward_backward(cpu(pc_in.pts_), cpu(pc_out.pts_), cpu(pc_in.local_frames_), cpu(pc_out.local_frames_),
                                  cpu(nbh.neighbors_), cpu(x), *(cpu(a) for a in args[2:5]), 1.0 / r, n_out / e, cpu(g))
    tol = TOLS[ops._precision]
    for got in (with_t, without_t):
        for u, v, name in zip(got, ref[1:], ("dX", "dA", "dbeta", "dW")):
            assert rel_err(u, v) < tol, (kind, name)
    assert params_only[0] is None and rel_err(params_only[3], ref[4]) < tol
    shp = geom.shape(c_in, c_out, 32)
    needs = _lib.load().se3conv_bwd_needs_t(C.byref(shp), 1)
    # a down-convolution of this size takes the edge-major feature gradient (no U): T is needed; the others do without
    assert needs == (1 if kind == "down" else 0)
    assert _lib.load().se3conv_bwd_needs_t(C.byref(shp), 0) == 1  # no feature gradient, no U
    if kind == "up":    # U (n_b rows) is the smaller tensor: the product reads it even though T is there -> identical results
        assert torch.equal(with_t[3], without_t[3])
