"""GPU: clouds whose coordinates contain -0.0 (mirrored clouds, planar clouds with z = -0.0).  The per-batch boxes are
reduced with integer atomics on an order-preserving view of the floats; -0.0f compares equal to zero but has the
bit pattern INT_MIN, so the branch must be taken on the sign bit (geometry.hip atomic_min_f / atomic_max_f).  Every
grid-based search (ball query beyond 2048 sources, grid kNN, grid sub-sampling) hangs off those boxes; all of them
are compared with the oracle, bit for bit."""
import pytest
import torch

from conftest import canon_edges
from oracle import se3conv_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd as amd
    return amd


def _clouds():
    g = torch.Generator().manual_seed(11)
    n = 6000
    out = {}
    # -0.0 as the ONLY value on an axis (a plane z = -0.0): the maximum must leave its -inf start value
    p = torch.rand(n, 3, generator=g)
    p[:, 2] = -0.0
    out["plane_z_negzero"] = p
    # -0.0 mixed with negative values: a minimum that is already negative must not be overwritten
    p = torch.rand(n, 3, generator=g) - 1.0          # all coordinates in [-1, 0)
    p[::7, 0] = -0.0
    p[::5, 1] = -0.0
    out["negative_with_negzero"] = p
    # mirrored cloud with exact zeros of both signs
    q = torch.rand(n // 2, 3, generator=g)
    q[::3, 0] = 0.0
    p = torch.cat([q, q * torch.tensor([-1.0, 1.0, 1.0])])   # 0.0 * -1 = -0.0
    out["mirrored"] = p
    return out


@pytest.mark.parametrize("name", ["plane_z_negzero", "negative_with_negzero", "mirrored"])
def test_boxes_ball_query_subsample_knn_with_negative_zero(amd, name):
    pts = _clouds()[name]
    assert bool((pts == 0).any()) and bool(torch.signbit(pts[pts == 0]).any()), "the case must contain -0.0"
    n = pts.shape[0]
    bid = torch.sort(torch.randint(0, 2, (n,), generator=torch.Generator().manual_seed(3), dtype=torch.int32)).values
    bid[-1] = 1
    # boxes: exact minima / maxima per batch element (values compare equal; the sign of a zero is free)
    mn, mx = amd.ops.batch_aabb(pts.to(DEV), bid.to(DEV), 2)
    for b in range(2):
        sel = pts[bid == b]
        assert torch.equal(mn[b].cpu(), sel.min(0).values) and torch.equal(mx[b].cpu(), sel.max(0).values)
        assert torch.isfinite(mn[b]).all() and torch.isfinite(mx[b]).all()
    # ball query through the cell grid (n_src > 2048)
    r = 0.07
    nb_r, ends_r = O.ball_query(pts, pts, bid, bid, r)
    nb, ends = amd.ops.ball_query(pts.to(DEV), pts.to(DEV), bid.to(DEV), bid.to(DEV), r, 2)
    assert torch.equal(ends.cpu(), ends_r)
    assert torch.equal(canon_edges(nb), canon_edges(nb_r))
    # grid sub-sampling: cell ids and level size
    ids_r, n_r, _pts_r, bid_r = O.grid_subsample(pts, bid, 0.1)
    cells = amd.ops.grid_subsample(pts.to(DEV), bid.to(DEV), 0.1, 2)
    assert cells.n_cells == n_r
    assert torch.equal(cells.cell_ids.cpu().to(torch.int64), ids_r.to(torch.int64))
    assert torch.equal(cells.batch_ids.cpu().to(torch.int64), bid_r.to(torch.int64))
    # kNN through the cell grid equals the all-pairs scan
    k_grid = amd.ops.knn_query(pts.to(DEV), bid.to(DEV), 8, 2, "grid")
    k_scan = amd.ops.knn_query(pts.to(DEV), bid.to(DEV), 8, 2, "scan")
    assert torch.equal(k_grid, k_scan)
