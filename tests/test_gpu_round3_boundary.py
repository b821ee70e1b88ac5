"""GPU, round 3: the pieces of the f-2 / f-1 boundary the task scripts call around the hot path --
random one-point-per-cell grid sub-sampling (GridSubSample(..., p_rnd_sample=True), pc/GridSubSample.py:43-54, 66-67,
83-91; tasks/SemSeg/train_dfaust_rot.py:143-149), the sub-sample method dispatch of PointHierarchy
(pc/PointHierarchy.py:46-52), k-NN neighbourhoods up to k = 64 / between two clouds (pc/KnnNeighborhood.py:38-84) and the
'matrix' / 'quaternion' relative-rotation descriptors (pc/RotationFunctions.py:593-600).

Integer work (ids, picks, labels, batch ids, neighbour indices) is compared bit-exactly; fp32 layer outputs against the
reference fixtures at 2e-5 (they run through the materialised fp32 path)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz, rel_err
from oracle import se3conv_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd as amd
    return amd


# ------------------------------------------------------------------------------------- random grid sub-sample
def test_grid_rnd_matches_reference_fixture(amd):
    d = load_npz(os.path.join(GOLDEN, "grid_rnd.npz"), DEV)
    pc = amd.pc.Pointcloud(d["pts"], d["batch"])
    samp = amd.pc.GridSubSample(pc, float(d["cell"]), p_rnd_sample=True, p_rnd_values=d["u"])
    # integer work, bit-exact: the cell of every point, and the reference's `ids_` from the same uniform numbers
    assert torch.equal(samp.cell_ids_, d["cell_ids"])
    assert torch.equal(samp.ids_, d["ids"])
    n_cells = d["u"].shape[0]
    # the represented point is a member of its cell (WHICH member sits at a position of the cell-sorted list is left
    # open by the reference: its argsort is not stable; here it is input order)
    assert torch.equal(samp.cell_ids_[samp.picked_.long()].long(), torch.arange(n_cells, device=DEV))
    assert torch.equal(samp.sorted_ids_[samp.ids_.long()], samp.picked_)
    assert torch.equal(samp.__subsample_tensor__(pc.pts_, "avg"), pc.pts_[samp.picked_.long()])
    # the gather / scatter maps with the reference's own cell-sorted list: values and gradients bit-exact
    samp.picked_ = amd.ops.rows_gather(d["sorted_ids"], samp.ids_)
    assert torch.equal(samp.picked_, d["picked"])
    assert torch.equal(samp.__subsample_tensor__(pc.pts_, "avg"), d["sub_pts"])
    assert torch.equal(samp.__subsample_tensor__(pc.batch_ids_, "max"), d["sub_batch"])
    lab = samp.__subsample_tensor__(d["labels"], "max")
    assert lab.dtype == torch.int64 and torch.equal(lab, d["sub_labels"])
    x = d["x"].clone().requires_grad_(True)
    y = samp.__subsample_tensor__(x, "avg")
    y.backward(d["sub_g"])
    assert torch.equal(y.detach(), d["sub_x"]) and torch.equal(x.grad, d["sub_dx"])
    z = d["z"].clone().requires_grad_(True)
    up = samp.__upsample_tensor__(z)
    up.backward(d["up_g"])
    assert torch.equal(up.detach(), d["up_y"]) and torch.equal(z.grad, d["up_dz"])


@pytest.mark.parametrize("n,batches,cell,seed", [(40000, 1, 0.04, 0), (30000, 5, 0.05, 1), (3, 1, 10.0, 2)])
def test_grid_rnd_matches_oracle_on_random_clouds(amd, n, batches, cell, seed):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(n, 3, generator=g)
    bid = torch.sort(torch.randint(0, batches, (n,), generator=g, dtype=torch.int32)).values
    bid[-1] = batches - 1
    cell_ids, n_cells, _, _ = O.grid_subsample(pts, bid, cell)
    u = torch.rand(n_cells, generator=g)
    u[0] = 0.0
    u[-1] = 1.0 - 2.0 ** -24  # the largest fp32 below 1: u * count may round up to count (clamped to the cell's last point)
    sorted_ids, ids, picked = O.grid_subsample_rnd(cell_ids, u)
    pc = amd.pc.Pointcloud(pts.to(DEV), bid.to(DEV))
    samp = amd.pc.GridSubSample(pc, cell, True, p_rnd_values=u.to(DEV))
    assert torch.equal(samp.sorted_ids_.cpu().long(), sorted_ids)  # stable: input order inside a cell, as the oracle
    assert torch.equal(samp.ids_.cpu().long(), ids) and torch.equal(samp.picked_.cpu().long(), picked)
    x = torch.randn(n_cells, 5, generator=g)
    assert torch.equal(samp.__upsample_tensor__(x.to(DEV)).cpu(), O.rows_upsample_rnd(x, picked, n))
    # default: the numbers are drawn on the device -- one member of every cell, no host round trip needed for them
    samp2 = amd.pc.GridSubSample(pc, cell, True)
    assert torch.equal(samp2.cell_ids_[samp2.picked_.long()].cpu().long(), torch.arange(n_cells))


def test_hierarchy_sub_sample_method_dispatch(amd):
    d = load_npz(os.path.join(GOLDEN, "grid_rnd.npz"), DEV)
    pc = amd.pc.Pointcloud(d["pts"], d["batch"])
    cells = [float(c) for c in d["h_cells"]]
    hier = amd.pc.PointHierarchy(pc, 2, "grid_rnd", grid_radii=cells)
    assert all(s.rnd_sample_ for s in hier.sub_sampled_objs_)
    # level 1: as many points as the reference's level, the same batch ids (a cell never spans batch elements), every
    # point a member of the source cloud (a random representative, not an average)
    assert hier.pcs_[1].pts_.shape == d["h_pts1"].shape
    assert torch.equal(hier.pcs_[1].batch_ids_, d["h_batch1"])
    picked = hier.sub_sampled_objs_[0].picked_.long()
    assert torch.equal(hier.pcs_[1].pts_, pc.pts_[picked])
    avg = amd.pc.PointHierarchy(pc, 2, "grid_avg", grid_radii=cells)
    assert not any(s.rnd_sample_ for s in avg.sub_sampled_objs_) and avg.pcs_[1].pts_.shape == hier.pcs_[1].pts_.shape
    with pytest.raises(NotImplementedError, match="farthest-point"):
        amd.pc.PointHierarchy(pc, 1, "fps", fps_ratios=[0.5])
    with pytest.raises(ValueError, match="unknown sub-sample method"):
        amd.pc.PointHierarchy(pc, 1, "grid_max", grid_radii=cells)


def test_create_hierarchy_call_sequence_of_the_task_script(amd):
    """tasks/SemSeg/train_dfaust_rot.py:108-158 (`create_hierarchy`, p_init_subsample=True with `output_subsample`)
    against `amd.pc`, call for call."""
    g = torch.Generator().manual_seed(4)
    n_b, per = 4, 2048
    pts = torch.rand(n_b * per, 3, generator=g).to(DEV)
    bid = torch.arange(n_b, dtype=torch.int32).repeat_interleave(per).to(DEV)
    feats = torch.ones(n_b * per, 1, device=DEV)
    labels = torch.randint(0, 24, (n_b * per,), generator=g).to(DEV)
    model = {"init_subsample": 0.04, "grid_subsamples": [0.08, 0.16, 0.32], "output_subsample": 0.02,
             "RefFrames": {"pca": True, "n_frames": 2, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}}
    with torch.no_grad():
        pc = amd.pc.Pointcloud(pts, bid)
        samp = amd.pc.GridSubSample(pc, model["init_subsample"])
        new_pts = samp.__subsample_tensor__(pc.pts_, "avg")
        new_bid = samp.__subsample_tensor__(pc.batch_ids_, "max")
        new_feat = samp.__subsample_tensor__(feats, "avg")
        new_pc = amd.pc.PointcloudRotEquiv(new_pts, new_bid, model["RefFrames"])
        hier = amd.pc.PointHierarchyRotEquiv(new_pc, len(model["grid_subsamples"]), "grid_avg", grid_radii=model["grid_subsamples"])
        levels_radii = [model["init_subsample"]] + model["grid_subsamples"]
        osamp = amd.pc.GridSubSample(pc, model["output_subsample"], p_rnd_sample=True)
        out_pts = osamp.__subsample_tensor__(pc.pts_, "avg")
        out_bid = osamp.__subsample_tensor__(pc.batch_ids_, "max")
        out_lab = osamp.__subsample_tensor__(labels, "max")
        out_pc = amd.pc.Pointcloud(out_pts, out_bid)
    assert len(hier.pcs_) == 4 and len(levels_radii) == 4
    assert new_feat.shape == (new_pts.shape[0], 1) and bool((new_feat == 1).all())
    sizes = [p.pts_.shape[0] for p in hier.pcs_]
    assert sizes == sorted(sizes, reverse=True) and sizes[0] == new_pts.shape[0]
    for p in hier.pcs_:
        assert p.local_frames_.shape == (p.pts_.shape[0], 2, 9)
    picked = osamp.picked_.long()
    assert out_lab.dtype == torch.int64 and torch.equal(out_lab, labels[picked]) and torch.equal(out_pts, pts[picked])
    assert torch.equal(out_bid, bid[picked]) and out_pc.num_batches() == n_b
    # the convolution towards the output cloud (FPNSegUNet.SEG_CONV_, models/FPNSegUNet.py:147-195) takes this pair
    nbh = amd.pc.BQNeighborhood(hier.pcs_[0], out_pc, 2.0 * levels_radii[0])
    assert nbh.start_ids_.shape[0] == out_pts.shape[0]


# ------------------------------------------------------------------------------------------------------ k-NN
@pytest.mark.parametrize("k", [33, 40, 64])
def test_knn_up_to_64_matches_oracle(amd, k):
    g = torch.Generator().manual_seed(k)
    pts = torch.rand(1500, 3, generator=g)
    bid = torch.sort(torch.randint(0, 3, (1500,), generator=g, dtype=torch.int32)).values
    bid[-20:] = 3  # a batch element with fewer than k points: -1 padding
    got = amd.ops.knn_query(pts.to(DEV), bid.to(DEV), k)
    assert torch.equal(got.cpu(), O.knn_query(pts, bid, k))
    pc = amd.pc.Pointcloud(pts.to(DEV), bid.to(DEV))
    nbh = amd.pc.KnnNeighborhood(pc, pc, k)
    ref = O.knn_query(pts, bid, k)
    assert nbh.neighbors_.shape[0] == int((ref >= 0).sum()) and int(nbh.start_ids_[-1]) == nbh.neighbors_.shape[0]
    with pytest.raises(NotImplementedError, match="at most 64"):
        amd.pc.KnnNeighborhood(pc, pc, 65)


@pytest.mark.parametrize("k", [1, 8, 16, 32, 64])
def test_knn_between_two_clouds_matches_oracle(amd, k):
    g = torch.Generator().manual_seed(100 + k)
    src = torch.rand(2500, 3, generator=g)
    bs = torch.sort(torch.randint(0, 4, (2500,), generator=g, dtype=torch.int32)).values
    bs[bs == 2] = 1  # batch element 2 has no source points
    q = torch.rand(700, 3, generator=g)
    bq = torch.sort(torch.randint(0, 4, (700,), generator=g, dtype=torch.int32)).values
    ref = O.knn_query_pair(src, bs, q, bq, k)
    got = amd.ops.knn_query_pair(src.to(DEV), bs.to(DEV), q.to(DEV), bq.to(DEV), k)
    assert torch.equal(got.cpu(), ref)
    pc_s, pc_q = amd.pc.Pointcloud(src.to(DEV), bs.to(DEV)), amd.pc.Pointcloud(q.to(DEV), bq.to(DEV))
    nbh = amd.pc.KnnNeighborhood(pc_s, pc_q, k)
    keep = ref >= 0
    assert nbh.neighbors_.dtype == torch.int64
    assert torch.equal(nbh.neighbors_[:, 1].cpu(), ref[keep].long())
    assert torch.equal(nbh.neighbors_[:, 0].cpu(), torch.arange(700)[:, None].expand(-1, k)[keep])
    assert torch.equal(nbh.start_ids_.cpu(), torch.cumsum(keep.sum(1), 0).to(torch.int32))
    full = amd.pc.KnnNeighborhood(pc_s, pc_q, k, p_keep_empty=True)
    assert full.neighbors_.shape == (700 * k, 2) and torch.equal(full.neighbors_[:, 1].cpu(), ref.reshape(-1).long())


def test_hierarchy_creates_knn_neighbourhoods(amd):
    g = torch.Generator().manual_seed(9)
    pts = torch.rand(3000, 3, generator=g).to(DEV)
    bid = torch.zeros(3000, dtype=torch.int32, device=DEV)
    hier = amd.pc.PointHierarchy(amd.pc.Pointcloud(pts, bid), 1, "grid_avg", grid_radii=[0.1])
    same = hier.create_neighborhood(0, 0, "knn", neihg_k=8)   # the reference's spelling of the keyword
    assert same is hier.create_neighborhood(0, 0, "knn", neigh_k=8) and same.k_ == 8
    down = hier.create_neighborhood(0, 1, "knn", neihg_k=12)  # level-0 sources for level-1 samples
    ref = O.knn_query_pair(pts.cpu(), bid.cpu(), hier.pcs_[1].pts_.cpu(), hier.pcs_[1].batch_ids_.cpu(), 12)
    assert torch.equal(down.neighbors_[:, 1].cpu(), ref.reshape(-1).long())
    assert hier.create_neighborhood(0, 0, "ball_query", bq_radius=0.1) is hier.create_neighborhood(0, 0, "ball_query", bq_radius=0.1)
    with pytest.raises(ValueError, match="unknown neighbourhood method"):
        hier.create_neighborhood(0, 0, "radius", bq_radius=0.1)


# --------------------------------------------------------------------------- other relative-rotation descriptors
@pytest.mark.parametrize("rel_rot,dims", [("matrix", 12), ("quaternion", 7)])
def test_rel_rot_layer_matches_reference_fixture(amd, rel_rot, dims):
    d = load_npz(os.path.join(GOLDEN, f"rel_rot_{rel_rot}.npz"), DEV)
    pc = amd.pc.PointcloudRotEquiv.from_frames(d["pts"], d["batch"], d["frames"])
    nbh = amd.pc.BQNeighborhood(pc, pc, float(d["radius"]))
    try:
        conv = amd.PNEConvLayerRotEquivFactory(dims, 32, "mlp_gelu", rel_rot).create_conv_layer(8, 16).to(DEV)
        conv.load_state_dict({"proj_axes_": d["proj_axes"], "proj_biases_": d["proj_biases"], "conv_weights_": d["conv_weights"],
                              "norm_neigh_dist_": d["rho"], "norm_num_neighs_": d["nu"]})
        x = d["x"].clone().requires_grad_(True)
        out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh)
        out.backward(d["grad_out"])
        rt = amd.PNEConvLayerRotEquiv.get_rot_tenors(pc, pc, nbh, conv.norm_neigh_dist_)
    finally:
        amd.PNEConvLayerRotEquiv.rel_rot_type = "6D"  # class attribute, as in the reference
    for got, key in ((out, "out"), (x.grad, "dx"), (conv.proj_axes_.grad, "dA"), (conv.proj_biases_.grad, "dbeta"),
                     (conv.conv_weights_.grad, "dW")):
        assert rel_err(got, d[key]) < 2e-5, key
    assert rt["rel_pts_rel_orient"].shape[1] == dims and torch.equal(rt["neighbs_start_ids"], d["rt_ends"])
    ref_nb, nb = d["rt_neighbs"].long(), rt["neighbs"]
    big = int(ref_nb[:, 1].max()) + 1
    o_ref, o_new = torch.argsort(ref_nb[:, 0] * big + ref_nb[:, 1]), torch.argsort(nb[:, 0] * big + nb[:, 1])
    assert torch.equal(ref_nb[o_ref], nb[o_new])
    assert rel_err(rt["rel_pts_rel_orient"][o_new], d["rt_desc"][o_ref]) < 2e-6
    with pytest.raises(ValueError, match="descriptor has"):
        bad = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu", rel_rot).create_conv_layer(8, 16).to(DEV)
        try:
            bad(p_pc_in=pc, p_pc_out=pc, p_in_features=d["x"], p_neighborhood=nbh)
        finally:
            amd.PNEConvLayerRotEquiv.rel_rot_type = "6D"
