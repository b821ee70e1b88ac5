"""Scope row f-1: self-kNN + PCA reference frames.  CPU: oracle vs the fixture generated from the reference's
PointcloudRotEquiv(pca=True).  GPU: the HIP kernels vs oracle / fixture.

Eigenvector signs are implementation-defined (LAPACK), so frames are compared as per-point SETS: the four
sign-flipped copies of the free case are invariant under that choice; in the fixed-axis case the reference's
up-axis column can come out as -e_axis, ours is +e_axis by construction (documented), so the in-plane axes are
compared up to sign.  Points whose covariance has nearly equal eigenvalues are skipped for the vector
comparison (ill-conditioned eigenvectors) but not for orthonormality / handedness."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_npz
from oracle import se3conv_oracle as O

DEV = "cuda:0"


def frame_sets_match(a, b, tol):
    """a, b: [N,P,9]; every frame of a has a partner in b within tol (max-abs), and vice versa."""
    d = (a[:, :, None, :] - b[:, None, :, :]).abs().amax(-1)  # [N,P,P]
    return (d.amin(2) < tol).all(1) & (d.amin(1) < tol).all(1)


def well_conditioned(pts, knn, axis=None, gap=5e-2):
    ids = knn.long().clone()
    rows = torch.arange(ids.shape[0])[:, None].expand_as(ids)
    ids[ids < 0] = rows[ids < 0]
    nm = pts[ids].clone()
    if axis:
        nm[:, :, axis] = 0
    nm = nm - nm.mean(1, keepdim=True)
    ev = torch.linalg.eigvalsh(torch.einsum("bij,bjk->bik", nm.transpose(1, 2), nm))
    ev = ev[:, 1:] if axis else ev  # the zeroed coordinate gives an exact 0 that is always separated
    rel = (ev[:, 1:] - ev[:, :-1]) / ev[:, -1:].clamp_min(1e-12)
    return rel.amin(1) > gap


def test_oracle_knn_and_pca_match_reference_fixture():
    d = load_npz(os.path.join(GOLDEN, "pca_frames.npz"))
    knn = O.knn_query(d["pts"], d["batch"], 16)
    assert torch.equal(knn, d["knn"])
    assert bool((knn[:, 0] == torch.arange(knn.shape[0])).all())  # the point itself comes first
    assert bool((d["batch"][knn.long()] == d["batch"][:, None]).all())
    for tag, axis in (("free", False), ("axis2", 2), ("axis1", 1)):
        fr = O.sample_reference_frames_pca(d["pts"], knn, axis)
        assert fr.shape == d[f"frames_{tag}"].shape
        assert float((fr - d[f"frames_{tag}"]).abs().max()) < 1e-5


@pytest.mark.gpu
def test_gpu_knn_matches_oracle(built_library):
    import se3conv3d_amd as amd

    g = torch.Generator().manual_seed(3)
    pts = torch.rand(3000, 3, generator=g)
    bid = torch.sort(torch.randint(0, 3, (3000,), generator=g, dtype=torch.int32)).values
    bid[-5:] = 3  # a batch element with fewer than k points -> -1 padding
    for k in (8, 16, 20):
        ref = O.knn_query(pts, bid, k)
        got = amd.ops.knn_query(pts.to(DEV), bid.to(DEV), k).cpu()
        assert torch.equal(got, ref)
    assert int((ref[-1] < 0).sum()) == 20 - 5
    nbh = amd.pc.KnnNeighborhood.__new__(amd.pc.KnnNeighborhood)
    pc = amd.pc.Pointcloud(pts.to(DEV), bid.to(DEV))
    amd.pc.KnnNeighborhood.__init__(nbh, pc, pc, 16, p_keep_empty=True)
    assert nbh.neighbors_.shape == (3000 * 16, 2) and torch.equal(nbh.start_ids_.cpu(), (torch.arange(3000) + 1).int() * 16)


@pytest.mark.gpu
def test_gpu_knn_grid_equals_scan(built_library):
    """The cell-grid search must return exactly what the all-pairs scan returns (same order, same ties, same -1
    padding), whatever the cloud looks like: the cell size is only a speed knob."""
    import se3conv3d_amd as amd

    g = torch.Generator().manual_seed(11)
    clouds = {}
    clouds["cube"] = (torch.rand(20000, 3, generator=g), torch.zeros(20000, dtype=torch.int32))
    v = torch.randn(15000, 3, generator=g)
    clouds["sphere_shell"] = (v / v.norm(dim=1, keepdim=True), torch.zeros(15000, dtype=torch.int32))
    flat = torch.rand(12000, 3, generator=g)
    flat[:, 2] = 0.25
    clouds["plane"] = (flat, torch.zeros(12000, dtype=torch.int32))
    line = torch.zeros(9000, 3)
    line[:, 0] = torch.rand(9000, generator=g)
    clouds["line"] = (line, torch.zeros(9000, dtype=torch.int32))
    # batch elements of very different size and density, the last one smaller than k
    sizes = [9000, 300, 4000, 7]
    pts = torch.cat([torch.rand(m, 3, generator=g) * (1.0 + 3.0 * i) for i, m in enumerate(sizes)])
    bid = torch.cat([torch.full((m,), i, dtype=torch.int32) for i, m in enumerate(sizes)])
    clouds["ragged_batches"] = (pts, bid)
    dup = torch.rand(5000, 3, generator=g)
    dup = torch.cat([dup, dup[:3000]])  # exact duplicates: ties at distance 0 and beyond
    clouds["duplicates"] = (dup, torch.zeros(8000, dtype=torch.int32))
    clumps = torch.cat([torch.rand(50, 3, generator=g) * 100.0,  # sparse far-away points force the fallback
                        torch.rand(9000, 3, generator=g) * 0.01])
    clouds["clumped"] = (clumps, torch.zeros(9050, dtype=torch.int32))
    for name, (p, b) in clouds.items():
        for k in (8, 16):
            scan = amd.ops.knn_query(p.to(DEV), b.to(DEV), k, method="scan")
            grid = amd.ops.knn_query(p.to(DEV), b.to(DEV), k, method="grid")
            assert torch.equal(scan, grid), f"{name}, k={k}: {int((scan != grid).any(dim=1).sum())} rows differ"
    ref = O.knn_query(clouds["ragged_batches"][0], clouds["ragged_batches"][1], 16)
    assert torch.equal(amd.ops.knn_query(pts.to(DEV), bid.to(DEV), 16, method="grid").cpu(), ref)


@pytest.mark.gpu
def test_gpu_pca_frames_match_reference_fixture(built_library):
    import se3conv3d_amd as amd

    d = load_npz(os.path.join(GOLDEN, "pca_frames.npz"))
    pts, knn = d["pts"], d["knn"]
    for tag, axis in (("free", None), ("axis2", 2), ("axis1", 1)):
        fr = amd.ops.pca_frames(pts.to(DEV), knn.to(DEV), axis).cpu()
        gold = d[f"frames_{tag}"]
        assert fr.shape == gold.shape
        m = fr.reshape(fr.shape[0], fr.shape[1], 3, 3)
        eye = torch.eye(3).expand_as(m)
        assert float((m.transpose(2, 3) @ m - eye).abs().max()) < 1e-5          # orthonormal columns
        # right-handed -- except fixed_axis = 1, where the reference permutes columns [0, 2, 1] AFTER its orientation
        # fix (RotationFunctions.py:400-401) and so returns left-handed frames; reproduced as is (quirk 7)
        hand = -1.0 if axis == 1 else 1.0
        assert float((torch.linalg.det(m) - hand).abs().max()) < 1e-5
        assert float((torch.linalg.det(gold.reshape(m.shape)) - hand).abs().max()) < 1e-5
        ok = well_conditioned(pts, knn, axis)
        assert int(ok.sum()) > 0.7 * ok.shape[0]
        if axis is None:
            assert bool(frame_sets_match(fr, gold, 2e-3)[ok].all())
        else:
            up = 2  # the fixed axis is the third basis vector (after the reference's column permutation for axis 1)
            col = {2: 2, 1: 1}[axis]
            assert float((m[:, :, :, col].abs() - torch.eye(3)[axis]).abs().max()) < 1e-6 and bool((m[:, :, axis, col] > 0).all())
            g = gold.reshape(gold.shape[0], gold.shape[1], 3, 3)
            other = [c for c in range(3) if c != col]
            for c in other:  # in-plane axes agree with the reference up to sign
                dots = (m[:, 0, :, c] * g[:, 0, :, c]).sum(-1).abs()
                assert bool((dots[ok] > 1 - 1e-4).all())
            # the second frame is the first rotated by pi about the fixed axis
            assert float((m[:, 1, :, other] + m[:, 0, :, other]).abs().max()) < 1e-6


@pytest.mark.gpu
def test_pointcloud_with_pca_frames_runs_the_layer(built_library):
    """PointcloudRotEquiv(pca=True) end to end on the GPU: cached 'se3-all' frames, random permutation per point,
    n_frames kept, and the conv consumes them (rotation invariance holds for PCA frames of the rotated cloud)."""
    import se3conv3d_amd as amd

    torch.manual_seed(4)
    n = 4000
    pts = torch.rand(n, 3, device=DEV)
    bid = torch.zeros(n, dtype=torch.int32, device=DEV)
    cfg = {"pca": True, "n_frames": 2, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
    pc = amd.pc.PointcloudRotEquiv(pts, bid, cfg)
    assert pc.local_frames_.shape == (n, 2, 9) and pc.n_frames_ == 2
    allf = pc.local_frames_pca_cache_["se3-all"]
    assert allf.shape == (n, 4, 9)
    # the two kept frames are two different members of the point's four
    d = (pc.local_frames_[:, :, None, :] - allf[:, None, :, :]).abs().amax(-1)
    assert bool((d.amin(2) == 0).all()) and bool((d.argmin(2)[:, 0] != d.argmin(2)[:, 1]).all())
    nbh = amd.pc.BQNeighborhood(pc, pc, 0.08)
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(32, 32).to(DEV)
    conv.norm_neigh_dist_.fill_(1 / 0.08)
    conv.norm_num_neighs_.fill_(n / nbh.neighbors_.shape[0])
    x = torch.randn(n * 2, 32, device=DEV)
    with torch.no_grad():
        out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh)
    assert out.shape == (n * 2, 32) and bool(torch.isfinite(out).all())


@pytest.mark.gpu
def test_global_frames_from_ref_frames_pts_and_standard_knn(built_library):
    """PointcloudRotEquiv(ref_frames_pts=...) (pc/PointcloudRotEquiv.py:80-128: one point per batch element, frames
    from the PCA of the whole element, RotationFunctions.py:265-304) against a direct eigen-decomposition of the
    element's scatter matrix, and `standard_knn=True` (test_scannet_rot.py:110) giving the frames of the default."""
    import se3conv3d_amd as amd

    torch.manual_seed(6)
    b, m = 5, 700
    ref = torch.randn(b, m, 3) * torch.tensor([3.0, 1.5, 0.5]) @ O.quaternion_to_matrix(
        torch.nn.functional.normalize(torch.randn(4), dim=0)).t() + torch.randn(b, 1, 3)
    centres = ref.mean(1)
    cfg = {"pca": True, "n_frames": 4, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
    pc = amd.pc.PointcloudRotEquiv(centres.to(DEV), torch.arange(b, dtype=torch.int32, device=DEV), cfg,
                                   ref_frames_pts=ref.reshape(-1, 3).to(DEV))
    assert pc.local_frames_.shape == (b, 4, 9) and pc.batch_ids_considering_frames_.shape == (b * 4,)
    allf = pc.local_frames_pca_cache_["se3-all"].cpu().reshape(b, 4, 3, 3)
    pcen = (ref - ref.mean(1, keepdim=True)).double()
    _, vec = torch.linalg.eigh(pcen.transpose(1, 2) @ pcen)          # ascending eigenvalues, columns = axes
    for i in range(b):
        f0 = allf[i, 0].double()
        assert abs(float(torch.linalg.det(f0)) - 1.0) < 1e-5
        for c in range(3):                                            # every axis up to sign
            assert abs(float((f0[:, c] * vec[i][:, c]).sum())) > 1 - 1e-4
        signs = torch.tensor([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=torch.double)
        for k in range(4):                                            # the four proper sign flips, in the reference's order
            assert float((allf[i, k].double() - f0 * signs[k][None, :]).abs().max()) < 1e-6
    # the kept frames are a permutation of the four
    d = (pc.local_frames_.cpu()[:, :, None, :] - allf.reshape(b, 1, 4, 9)).abs().amax(-1)
    assert bool((d.amin(2) == 0).all()) and sorted(d.argmin(2)[0].tolist()) == [0, 1, 2, 3]
    with pytest.raises(NotImplementedError):
        amd.pc.PointcloudRotEquiv(centres.to(DEV), torch.arange(b, dtype=torch.int32, device=DEV), dict(cfg, fixed_axis=2),
                                  ref_frames_pts=ref.reshape(-1, 3).to(DEV))
    rnd = amd.pc.PointcloudRotEquiv(centres.to(DEV), torch.arange(b, dtype=torch.int32, device=DEV),
                                    {"pca": False, "n_frames": 2, "fixed_axis": False}, ref_frames_pts=ref.reshape(-1, 3).to(DEV))
    assert rnd.local_frames_.shape == (1, 2, 9)                       # the reference samples for n_origins = 1 here
    # standard_knn: same exact neighbour sets, hence the same PCA frames
    n = 3000
    pts = torch.rand(n, 3, device=DEV)
    bid = torch.zeros(n, dtype=torch.int32, device=DEV)
    a = amd.pc.PointcloudRotEquiv(pts, bid, cfg)
    s = amd.pc.PointcloudRotEquiv(pts, bid, cfg, standard_knn=True)
    assert torch.equal(a.local_frames_pca_cache_["se3-all"], s.local_frames_pca_cache_["se3-all"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_all,n_frames", [(5000, 4, 2), (3000, 4, 4), (777, 2, 1), (1, 4, 1), (0, 4, 2)])
def test_shuffle_frames_is_the_order_of_the_draws(built_library, n, n_all, n_frames):
    """se3_shuffle_frames against numpy: out[p, j] = all[p, argsort(draws[p])[j]] (stable: ties to the lower index), i.e.
    the multinomial-without-replacement + gather of PointcloudRotEquiv.py:100-117, 146-167 with the draws given."""
    import se3conv3d_amd as amd

    g = torch.Generator().manual_seed(n + n_all)
    allf = torch.randn(n, n_all, 9, generator=g)
    draws = torch.rand(n, n_all, generator=g)
    if n > 10:
        draws[3] = 0.25          # all equal: input order
        draws[4, -1] = draws[4, 0]
    out = amd.ops.shuffle_frames(allf.to(DEV), n_frames, draws.to(DEV)).cpu()
    perm = np.argsort(draws.numpy(), axis=1, kind="stable")[:, :n_frames]
    want = np.take_along_axis(allf.numpy(), perm[:, :, None], axis=1)
    assert out.shape == (n, n_frames, 9) and np.array_equal(out.numpy(), want)
    if n >= 3000:  # the device's own draws: every frame position is taken about equally often
        own = amd.ops.shuffle_frames(allf.to(DEV), n_frames).cpu()
        first = (own[:, 0, None, :] == allf).all(-1).float().mean(0)
        assert float((first - 1.0 / n_all).abs().max()) < 0.05
    with pytest.raises(ValueError):
        amd.ops.shuffle_frames(allf.to(DEV), n_frames, draws[:, :1].to(DEV))


@pytest.mark.gpu
def test_frame_neighbourhood_is_built_from_the_id_table_on_demand(built_library):
    """A PCA-frame cloud keeps only the [N, k] id table of its k-NN; get_ref_frame_neighborhood builds the reference's
    neighbourhood object (PointcloudRotEquiv.py:54-75) from that table when asked, and a hierarchy level inherits its
    parent's batch count (no read-back), several batch elements of uneven size included."""
    import se3conv3d_amd as amd

    torch.manual_seed(8)
    sizes = [700, 1, 2300, 64, 935]
    pts = torch.cat([torch.rand(s, 3) + 2.0 * i for i, s in enumerate(sizes)]).to(DEV)
    bid = torch.cat([torch.full((s,), i, dtype=torch.int32) for i, s in enumerate(sizes)]).to(DEV)
    cfg = {"pca": True, "n_frames": 2, "fixed_axis": False, "neigh_method": "knn", "neigh_kwargs": {"neigh_k": 16}}
    pc = amd.pc.PointcloudRotEquiv(pts, bid, cfg)
    assert pc.neigh_cache_ == {}
    ids = pc._self_knn_ids(16)
    assert ids.shape == (4000, 16) and bool((ids[:, 0] == torch.arange(4000, device=DEV)).all())
    nbh = pc.get_ref_frame_neighborhood("knn", neigh_k=16)
    assert nbh.neighbors_.shape == (4000 * 16, 2)
    assert bool((nbh.neighbors_[:, 1].reshape(4000, 16) == ids).all())
    assert bool((nbh.neighbors_[:, 0].reshape(4000, 16) == torch.arange(4000, device=DEV)[:, None]).all())
    assert bool((nbh.start_ids_ == torch.arange(1, 4001, device=DEV) * 16).all())
    assert bool((ids[700] == torch.tensor([700] + [-1] * 15, device=DEV)).all())   # the one-point element: itself, then padding
    # boxes per batch element (64-point groups that straddle elements take the segmented path of batch_aabb_kernel)
    mn, mx = pc.aabb()
    for i, s in enumerate(sizes):
        sel = pts[bid == i]
        assert torch.equal(mn[i], sel.amin(0)) and torch.equal(mx[i], sel.amax(0))
    assert bool((pc.batch_ids_considering_frames_ == bid.repeat_interleave(2)).all())
    hier = amd.pc.PointHierarchyRotEquiv(pc, 2, "grid_avg", grid_radii=[0.1, 0.2])
    for lvl in hier.pcs_[1:]:
        assert lvl._num_batches == 5 and int(lvl.batch_size_) == 5
        assert lvl.local_frames_.shape == (lvl.pts_.shape[0], 2, 9)
