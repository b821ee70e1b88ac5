"""CPU: the oracle against the fixtures generated from the reference's Python (tools/gen_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, canon_edges, golden_layer_files, load_npz, rel_err
from oracle import se3conv_oracle as O

FILES = golden_layer_files()
TOL = 2e-6  # fp32 CPU restatement vs reference Python, ||d||/||ref||


def test_fixtures_present():
    assert len(FILES) >= 7
    assert os.path.exists(os.path.join(GOLDEN, "rotation_fns.npz"))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_layer_forward_backward_matches_reference(path):
    d = load_npz(path)
    out, dx, da, db, dw = O.conv_forward_backward(
        d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(), d["x"],
        d["proj_axes"], d["proj_biases"], d["conv_weights"], d["rho"], d["nu"], d["grad_out"])
    assert out.shape == d["out"].shape
    assert rel_err(out, d["out"]) < TOL
    assert rel_err(dx, d["dx"]) < TOL
    assert rel_err(da, d["dA"]) < TOL
    assert rel_err(db, d["dbeta"]) < TOL
    assert rel_err(dw, d["dW"]) < TOL


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_fp64_oracle_agrees(path):
    """The fp64 evaluation of the same formula bounds the fp32 reference's own rounding."""
    d = load_npz(path)
    out, dx, da, db, dw = O.conv_forward_backward(
        d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(), d["x"],
        d["proj_axes"], d["proj_biases"], d["conv_weights"], d["rho"], d["nu"], d["grad_out"], dtype=torch.float64)
    assert rel_err(d["out"], out) < TOL and rel_err(d["dx"], dx) < TOL and rel_err(d["dW"], dw) < TOL


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_rot_tensors_match_reference(path):
    d = load_npz(path)
    rt = O.get_rot_tensors(d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(), d["rho"])
    assert torch.equal(rt["neighbs_start_ids"], d["rt_ends"])
    # the reference's sort is unstable: compare per output row as multisets keyed by the source row
    ref_nb, nb = d["rt_neighbs"].long(), rt["neighbs"]
    assert torch.equal(ref_nb[:, 0], nb[:, 0])
    big = int(max(ref_nb[:, 1].max(), nb[:, 1].max())) + 1
    o_ref = torch.argsort(ref_nb[:, 0] * big + ref_nb[:, 1])
    o_new = torch.argsort(nb[:, 0] * big + nb[:, 1])
    assert torch.equal(ref_nb[o_ref], nb[o_new])
    assert rel_err(rt["rel_pts_rel_orient"][o_new], d["rt_desc"][o_ref]) < TOL


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_ball_query_matches_reference_edges(path):
    d = load_npz(path)
    nb, ends = O.ball_query(d["pts_in"], d["pts_out"], d["batch_in"], d["batch_out"], float(d["radius"]))
    assert torch.equal(ends, d["ends"])
    assert torch.equal(canon_edges(nb), canon_edges(d["neighbors"]))


@pytest.mark.parametrize("name", ["layer_down_n512_n128_f2.npz", "layer_sparse_n200_f2.npz", "layer_n256_f2_c64.npz"])
def test_grid_search_loses_no_neighbour(name):
    """The reference's grid windows (3x3 pencils x [z-1,z+1]) give the brute-force edge set."""
    d = load_npz(os.path.join(GOLDEN, name))
    nb_g, ends_g = O.ball_query_grid(d["pts_in"], d["pts_out"], d["batch_in"], d["batch_out"], float(d["radius"]))
    assert torch.equal(ends_g, d["ends"])
    assert torch.equal(canon_edges(nb_g), canon_edges(d["neighbors"]))


@pytest.mark.parametrize("path", FILES[:3], ids=[os.path.basename(f) for f in FILES[:3]])
def test_feat_basis_proj_and_grad(path):
    """feat_basis_proj / _grad restatements against autograd of the dense formula."""
    d = load_npz(path)
    rt = O.get_rot_tensors(d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(), d["rho"])
    phi = O.kernel_mlp(rt["rel_pts_rel_orient"], d["proj_axes"], d["proj_biases"]).requires_grad_(True)
    x = d["x"].clone().requires_grad_(True)
    t = O.feat_basis_proj(phi, x, rt["neighbs"], rt["neighbs_start_ids"])
    out = torch.einsum("nik,iko->no", t, d["conv_weights"]) / d["frames_in"].shape[1] * d["nu"]
    assert rel_err(out, d["out"]) < TOL
    g = torch.randn_like(t)
    t.backward(g)
    g_feat, g_basis = O.feat_basis_proj_grad(phi.detach(), x.detach(), rt["neighbs"], rt["neighbs_start_ids"], g)
    assert rel_err(g_feat, x.grad) < TOL and rel_err(g_basis, phi.grad) < TOL


def test_ema_trajectory():
    for path in FILES:
        d = load_npz(path)
        rho, nu = torch.tensor(0.0), torch.tensor(0.0)
        for step in range(d["ema"].shape[0]):
            rho, nu = O.ema_update(rho, nu, float(d["radius"]), d["ends"].shape[0], d["neighbors"].shape[0])
            assert abs(float(rho) - float(d["ema"][step, 0])) <= 1e-6 * abs(float(d["ema"][step, 0]))
            assert abs(float(nu) - float(d["ema"][step, 1])) <= 1e-6 * abs(float(d["ema"][step, 1]))


def test_rotation_functions():
    d = load_npz(os.path.join(GOLDEN, "rotation_fns.npz"))
    fa, fb = d["frames_a"], d["frames_b"]
    nb = torch.stack((torch.arange(5), torch.arange(5)), 1)
    desc = O.edge_descriptors(torch.zeros(5, 3) + d["dirs"], torch.zeros(5, 3), fb, fa, nb, torch.tensor(1.0))
    # descriptor = [dirs @ R_a, rows 0,1 of R_a^T R_b] in pair order a*F_b + b
    local = d["local"][:, :, None, :].expand(-1, -1, 4, -1).reshape(5, 8, 3)
    assert rel_err(desc[..., :3], local) < TOL
    assert rel_err(desc[..., 3:], d["rel6"]) < TOL
    assert rel_err(d["relmat"][..., :6], d["rel6"]) == 0.0
    combos = torch.tensor([(a, b) for a in range(2) for b in range(4)])
    assert torch.equal(combos, d["combos"])


def test_dropped_rows_quirk():
    """pad_rows=False reproduces the reference quirk: trailing output rows without edges vanish."""
    g = torch.Generator().manual_seed(0)
    pts_in = torch.rand(50, 3, generator=g)
    pts_out = torch.cat((torch.rand(20, 3, generator=g), torch.full((2, 3), 9.0)))  # last 2 samples isolated
    fi, fo = O.random_frames(50, 2, g), O.random_frames(22, 2, g)
    z = torch.zeros(50, dtype=torch.int32), torch.zeros(22, dtype=torch.int32)
    nb, ends = O.ball_query(pts_in, pts_out, z[0], z[1], 0.4)
    a, b, w = O.init_parameters(9, 8, 8, 32, g)
    x = torch.randn(100, 8, generator=g)
    full = O.conv_forward(pts_in, pts_out, fi, fo, nb, x, a, b, w, torch.tensor(2.5), torch.tensor(0.1))
    ref_like = O.conv_forward(pts_in, pts_out, fi, fo, nb, x, a, b, w, torch.tensor(2.5), torch.tensor(0.1), pad_rows=False)
    assert full.shape[0] == 44 and ref_like.shape[0] < 44
    assert torch.equal(full[: ref_like.shape[0]], ref_like)
    assert float(full[ref_like.shape[0]:].abs().max()) == 0.0
    lean = O.conv_forward_edgewise(pts_in, pts_out, fi, fo, nb, x, a, b, w, torch.tensor(2.5), torch.tensor(0.1))
    assert rel_err(lean, full) < TOL


PNE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "pne_*.npz")))


@pytest.mark.parametrize("path", PNE_FILES, ids=[os.path.basename(f) for f in PNE_FILES])
def test_oracle_pne_layer_matches_reference_fixture(path):
    """Non-equivariant PNEConvLayer (scope row f-4): the restatement against the reference's own Python."""
    d = np.load(path)
    out, dx, da, db, dw = O.pne_conv_forward_backward(
        d["pts_in"], d["pts_out"], d["neighbors"], d["ends"], d["x"], d["proj_axes"], d["proj_biases"],
        d["conv_weights"], d["rho"], d["nu"], d["grad_out"])
    for got, key in ((out, "out"), (dx, "dx"), (da, "dA"), (db, "dbeta"), (dw, "dW")):
        ref = torch.as_tensor(d[key]).double()
        assert float((got - ref).norm() / ref.norm()) < 2e-6, key


def test_oracle_hierarchy_and_frame_pooling_match_reference_fixture():
    """Scope rows f-2 / f-3: grid sub-sampling (cell ids bit-exact, level points / batch ids), pool / up-sample maps
    and frame pooling with their gradients against the reference's PointHierarchy / feature_pooling."""
    d = np.load(os.path.join(GOLDEN, "hierarchy.npz"))
    pts, bid = torch.from_numpy(d["pts"]), torch.from_numpy(d["batch"])
    for lv in (0, 1):
        ids, m, lp, lb = O.grid_subsample(pts, bid, float(d["cells"][lv]))
        assert np.array_equal(ids.numpy(), d[f"cell_ids_l{lv}"].astype(np.int64))
        assert m == d[f"pts_l{lv + 1}"].shape[0]
        np.testing.assert_allclose(lp.numpy(), d[f"pts_l{lv + 1}"], rtol=0, atol=2e-7)
        assert np.array_equal(lb.numpy(), d[f"batch_l{lv + 1}"])
        pts, bid = lp, lb
    ids0 = torch.from_numpy(d["cell_ids_l0"]).long()
    m0 = d["pts_l1"].shape[0]
    for method in ("avg", "max"):
        x = torch.from_numpy(d[f"pool_{method}_x"]).requires_grad_(True)
        y = O.segment_pool(x, ids0, m0, method)
        y.backward(torch.from_numpy(d[f"pool_{method}_g"]))
        np.testing.assert_allclose(y.detach().numpy(), d[f"pool_{method}_y"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), d[f"pool_{method}_dx"], rtol=0, atol=1e-6)
    z = torch.from_numpy(d["up_z"]).requires_grad_(True)
    up = O.segment_upsample(z, ids0)
    up.backward(torch.from_numpy(d["up_g"]))
    assert np.array_equal(up.detach().numpy(), d["up_y"])
    np.testing.assert_allclose(z.grad.numpy(), d["up_dz"], rtol=0, atol=1e-5)
    f = int(d["frames"])
    for method in ("avg", "max", "min", "sum"):
        x = torch.from_numpy(d[f"fpool_{method}_x"]).requires_grad_(True)
        y = O.frame_pool(x, f, method)
        y.backward(torch.from_numpy(d[f"fpool_{method}_g"]))
        np.testing.assert_allclose(y.detach().numpy(), d[f"fpool_{method}_y"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(x.grad.numpy(), d[f"fpool_{method}_dx"], rtol=0, atol=1e-6)


def _load_block(d, blocks, factory):
    blk = blocks.ResNetFormer(32, 48, factory, blocks.BatchNormPC, 0.0)
    state = {k[len("state/"):]: torch.from_numpy(np.asarray(d[k])) for k in d.files if k.startswith("state/")}
    missing = blk.load_state_dict(state, strict=True)  # the reference's keys, one for one
    assert not missing.missing_keys and not missing.unexpected_keys
    return blk.train()


def test_resnetformer_block_glue_matches_reference_fixture():
    """The torch glue of se3conv3d_amd.blocks (norms, skips with their gains, point-wise MLP, linear skip) around a
    CPU stand-in of the convolution (the oracle) against the reference's ResNetFormer: the reference state_dict loads
    key for key, output, input gradient, every parameter gradient and the batch-norm running statistics agree."""
    import types
    from se3conv3d_amd import blocks

    d = np.load(os.path.join(GOLDEN, "resnetformer_block.npz"))
    pts, frames = torch.from_numpy(d["pts"]), torch.from_numpy(d["frames"])
    bid = torch.from_numpy(d["batch"])
    nb, _ = O.ball_query(pts, pts, bid, bid, float(d["radius"]))

    class OracleConv(torch.nn.Module):  # same parameter / buffer names as the layer
        def __init__(self, c_in, c_out):
            super().__init__()
            self.proj_axes_ = torch.nn.Parameter(torch.zeros(9, 32))
            self.proj_biases_ = torch.nn.Parameter(torch.zeros(32))
            self.conv_weights_ = torch.nn.Parameter(torch.zeros(c_in, 32, c_out))
            self.register_buffer("norm_neigh_dist_", torch.tensor(0.0))
            self.register_buffer("norm_num_neighs_", torch.tensor(0.0))

        def forward(self, p_pc_in, p_pc_out, p_in_features, p_neighborhood):
            return O.conv_forward(pts, pts, frames, frames, nb, p_in_features, self.proj_axes_, self.proj_biases_,
                                  self.conv_weights_, self.norm_neigh_dist_, self.norm_num_neighs_)

    factory = types.SimpleNamespace(create_conv_layer=lambda ci, co: OracleConv(ci, co))
    blk = _load_block(d, blocks, factory)
    pc = types.SimpleNamespace(batch_size_=torch.tensor(2), batch_ids_=bid)
    x = torch.from_numpy(d["x"]).requires_grad_(True)
    out = blk(pc, x, None)
    out.backward(torch.from_numpy(d["g"]))
    t = lambda k: torch.from_numpy(np.asarray(d[k]))
    assert rel_err(out, t("out")) < 5e-6 and rel_err(x.grad, t("dx")) < 5e-6
    for name, p in blk.named_parameters():
        assert rel_err(p.grad, t("grad/" + name)) < 2e-5, name
    for k, v in blk.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), d["after/" + k], rtol=1e-5, atol=1e-6)


# ---- round 3: random grid sub-sample, other relative-rotation descriptors, checkpoint key layout -------------------
def test_grid_rnd_oracle_matches_reference_fixture():
    """GridSubSample(..., p_rnd_sample=True) (GridSubSample.py:43-54): `ids_` from the stored uniform numbers is integer
    work (bit-exact); which point sits at a position of the cell-sorted list is left open by the reference (its argsort
    is not stable), so the gather / scatter maps are checked with the reference's own `sorted_ids_`."""
    d = load_npz(os.path.join(GOLDEN, "grid_rnd.npz"))
    cell_ids, n_cells, _, _ = O.grid_subsample(d["pts"], d["batch"], float(d["cell"]))
    assert torch.equal(cell_ids.to(torch.int32), d["cell_ids"]) and n_cells == d["u"].shape[0]
    sorted_ids, ids, picked = O.grid_subsample_rnd(cell_ids, d["u"])
    assert torch.equal(ids.to(torch.int32), d["ids"])
    # the oracle's pick and the reference's pick are points of the same cell
    assert torch.equal(cell_ids[picked], torch.arange(n_cells))
    assert torch.equal(d["cell_ids"].long()[d["picked"].long()], torch.arange(n_cells))
    ref_picked = d["sorted_ids"].long()[ids]
    assert torch.equal(ref_picked.to(torch.int32), d["picked"])
    assert torch.equal(d["pts"][ref_picked], d["sub_pts"]) and torch.equal(d["labels"][ref_picked], d["sub_labels"])
    assert torch.equal(d["batch"][ref_picked], d["sub_batch"])
    assert torch.equal(O.rows_upsample_rnd(d["z"], ref_picked, d["pts"].shape[0]), d["up_y"])
    assert torch.equal(O.rows_upsample_rnd(d["sub_g"], ref_picked, d["pts"].shape[0]), d["sub_dx"])  # gradient of the gather
    assert torch.equal(d["up_g"][ref_picked], d["up_dz"])                                             # gradient of the scatter


@pytest.mark.parametrize("rel_rot", ["matrix", "quaternion"])
def test_rel_rot_descriptors_and_layer_match_reference(rel_rot):
    """p_rel_rot = 'matrix' (D = 12) / 'quaternion' (D = 7), RotationFunctions.py:593-600: descriptors, forward, gradients."""
    d = load_npz(os.path.join(GOLDEN, f"rel_rot_{rel_rot}.npz"))
    nb = d["neighbors"].long()
    rt = O.get_rot_tensors(d["pts"], d["pts"], d["frames"], d["frames"], nb, d["rho"], rel_rot=rel_rot)
    assert torch.equal(rt["neighbs_start_ids"], d["rt_ends"])
    ref_nb, new_nb = d["rt_neighbs"].long(), rt["neighbs"]
    big = int(ref_nb[:, 1].max()) + 1
    o_ref, o_new = torch.argsort(ref_nb[:, 0] * big + ref_nb[:, 1]), torch.argsort(new_nb[:, 0] * big + new_nb[:, 1])
    assert torch.equal(ref_nb[o_ref], new_nb[o_new])
    assert rel_err(rt["rel_pts_rel_orient"][o_new], d["rt_desc"][o_ref]) < TOL
    out, dx, da, db, dw = O.conv_forward_backward(d["pts"], d["pts"], d["frames"], d["frames"], nb, d["x"], d["proj_axes"],
                                                  d["proj_biases"], d["conv_weights"], d["rho"], d["nu"], d["grad_out"],
                                                  rel_rot=rel_rot)
    for got, key in ((out, "out"), (dx, "dx"), (da, "dA"), (db, "dbeta"), (dw, "dW")):
        assert rel_err(got, d[key]) < TOL, key


def test_knn_pair_oracle_reduces_to_self_query():
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(300, 3, generator=g)
    bid = torch.sort(torch.randint(0, 3, (300,), generator=g, dtype=torch.int32)).values
    assert torch.equal(O.knn_query_pair(pts, bid, pts, bid, 9), O.knn_query(pts, bid, 9))
    q = torch.rand(40, 3, generator=g)
    qb = torch.sort(torch.randint(0, 4, (40,), generator=g, dtype=torch.int32)).values  # batch 3 has no sources
    ids = O.knn_query_pair(pts, bid, q, qb, 5)
    assert (ids[qb == 3] == -1).all() and (ids[qb < 3] >= 0).all()
    assert (bid[ids[qb < 3].long()] == qb[qb < 3][:, None]).all()
