"""GPU: the capacity-bounded ball query (se3_ball_query_bounded: no host round trip for the edge count, SURVEY.md
section 8 row a10 / the reference's syncs at ball_query.cu:46,49-50) -- same edge sets as the two-phase query, the
edge count and the overflow flag on the device, truncation that never lets a consumer read past the buffer, and a
neighbourhood + convolution step captured into one HIP graph."""
import pytest
import torch

from conftest import canon_edges, rel_err
from oracle import se3conv_oracle as O
from se3conv3d_amd.workloads import radius_for_degree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd as amd
    amd.set_precision("bf16x3")
    return amd


@pytest.mark.parametrize("n_src,n_dst,batches,r", [
    (6000, 1500, 3, 0.09),      # grid search, 64-bit keys (more than two batch elements)
    (6000, 1500, 2, 0.09),      # grid search, 32-bit keys with the fixed cell stride
    (5000, 5000, 1, 0.0007),    # 1400 cells per axis: indices beyond 1023 are clamped, the edge set must not change
    (9000, 9000, 1, 0.05),
    (1500, 2500, 2, 0.12),      # all-pairs search, offsets formed inside the store kernel
    (1500, 6000, 1, 0.12),      # all-pairs search, more samples than the inline prefix takes: scan launch
    (40, 500, 1, 0.4)])
def test_bounded_equals_two_phase_and_oracle(amd, n_src, n_dst, batches, r):
    g = torch.Generator().manual_seed(n_src)
    ps, pd = torch.rand(n_src, 3, generator=g), torch.rand(n_dst, 3, generator=g)
    if r < 0.01:
        pd = ps.clone()      # a cloud against itself: at least the self edges exist whatever the radius
    bs = torch.sort(torch.randint(0, batches, (n_src,), generator=g, dtype=torch.int32)).values
    bd = torch.sort(torch.randint(0, batches, (n_dst,), generator=g, dtype=torch.int32)).values
    bs[-1] = batches - 1
    nb_r, ends_r = O.ball_query(ps, pd, bs, bd, r)
    e = nb_r.shape[0]
    args = (ps.to(DEV), pd.to(DEV), bs.to(DEV), bd.to(DEV), r)
    nb, ends, info = amd.ops.ball_query_bounded(*args, capacity=e + 100, n_batches=batches)
    assert info.tolist() == [e, 0] and nb.shape == (e + 100, 2)
    assert torch.equal(ends.cpu(), ends_r) and torch.equal(canon_edges(nb[:e]), canon_edges(nb_r))
    nb2, ends2 = amd.ops.ball_query(*args, batches)
    assert torch.equal(nb[:e], nb2) and torch.equal(ends, ends2)          # same deterministic order as the two-phase call
    # exact fit, then too small: the flag is raised, the offsets are clamped, the head of the list is intact
    nb, ends, info = amd.ops.ball_query_bounded(*args, capacity=e, n_batches=batches)
    assert info.tolist() == [e, 0] and torch.equal(nb, nb2)
    cap = e // 2
    # the edge buffer sits inside a larger pre-filled arena: a truncated build must not write a row past its capacity
    arena = torch.full((cap + 64, 2), -7, dtype=torch.int32, device=DEV)
    nb, ends, info = amd.ops.ball_query_bounded(*args, capacity=cap, n_batches=batches, neighbors_out=arena[:cap])
    assert info.tolist() == [e, 1]
    assert torch.equal(ends.cpu(), torch.clamp(ends_r, max=cap)) and torch.equal(nb, nb2[:cap])
    assert nb.data_ptr() == arena.data_ptr() and bool((arena[cap:] == -7).all())
    # the optional dense list of source ids = column 1 of the edge list (the source-major list of a symmetric graph)
    nb, ends, info, src = amd.ops.ball_query_bounded(*args, capacity=e + 5, n_batches=batches, want_sources=True)
    assert torch.equal(src[:e], nb2[:, 1]) and torch.equal(nb[:e], nb2)
    nb, ends, info = amd.ops.ball_query_bounded(*args, capacity=0, n_batches=batches)
    assert info.tolist() == [e, 1 if e else 0] and int(ends.max()) == 0


def test_neighbourhood_and_conv_step_in_one_graph(amd):
    torch.manual_seed(0)
    n, f, c = 6000, 2, 64
    pc = amd.pc.PointcloudRotEquiv(torch.rand(n, 3, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV),
                                   {"pca": False, "n_frames": f, "fixed_axis": False})
    pc.num_batches()
    r = radius_for_degree(n, 24)
    ref_nbh = amd.pc.BQNeighborhood(pc, pc, r)
    e = ref_nbh.num_edges()
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c, c).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r), conv.norm_num_neighs_.fill_(n / e)
    x = torch.randn(n * f, c, device=DEV, requires_grad=True)
    g = torch.randn(n * f, c, device=DEV)
    out_ref = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh)
    out_ref.backward(g)
    dx_ref, dw_ref = x.grad.clone(), conv.conv_weights_.grad.clone()
    # drop the eager autograd graph: it was built on the default stream, and the AccumulateGrad nodes it keeps alive
    # would be reused -- with that stream -- by the captured backward (torch's capture rule: warm up on a side stream)
    out_ref = out_ref.detach().clone()

    holder = {}

    def step():
        x.grad = None
        conv.zero_grad(set_to_none=True)
        nbh = amd.pc.BQNeighborhood(pc, pc, r, p_capacity=int(e * 1.25))
        out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh)
        out.backward(g)
        holder.update(nbh=nbh, out=out.detach())   # no reference to the autograd graph survives the step

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):       # fails if anything on the path synchronises with the host
        step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    nbh = holder["nbh"]
    assert nbh.num_edges() == e and not nbh.overflowed() and nbh.neighbors_.shape[0] == int(e * 1.25)
    assert torch.equal(holder["out"], out_ref) and torch.equal(x.grad, dx_ref)
    assert torch.equal(conv.conv_weights_.grad, dw_ref)
    # a truncated neighbourhood still runs (results are those of the truncated graph) and says so
    small = amd.pc.BQNeighborhood(pc, pc, r, p_capacity=e // 3)
    assert small.overflowed() and small.num_edges() == e // 3
    out_small = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x.detach(), p_neighborhood=small)
    assert bool(torch.isfinite(out_small).all())


@pytest.mark.parametrize("poison", [False, True])
def test_bounded_neighbourhood_between_two_clouds_runs_forward_and_backward(amd, poison):
    """A capacity-bounded neighbourhood between DIFFERENT clouds: backward needs the source-major copy of the edge list,
    and the unset tail of the buffer must not reach the sort that builds it (se3_csr_transpose_bounded).  Outputs and
    every gradient equal the two-phase neighbourhood's bit for bit -- also when the tail holds in-range garbage ids."""
    torch.manual_seed(1)
    n_in, n_out, f, c_in, c_out = 5000, 1800, 2, 32, 64
    cfg = {"pca": False, "n_frames": f, "fixed_axis": False}
    pc_in = amd.pc.PointcloudRotEquiv(torch.rand(n_in, 3, device=DEV), torch.zeros(n_in, dtype=torch.int32, device=DEV), cfg)
    pc_out = amd.pc.PointcloudRotEquiv(torch.rand(n_out, 3, device=DEV), torch.zeros(n_out, dtype=torch.int32, device=DEV), cfg)
    pc_in.num_batches(), pc_out.num_batches()
    r = radius_for_degree(n_in, 20)
    ref_nbh = amd.pc.BQNeighborhood(pc_in, pc_out, r)
    e = ref_nbh.num_edges()
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c_in, c_out).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r), conv.norm_num_neighs_.fill_(n_out / e)
    x = torch.randn(n_in * f, c_in, device=DEV, requires_grad=True)
    g = torch.randn(n_out * f, c_out, device=DEV)

    def run(nbh):
        x.grad = None
        conv.zero_grad(set_to_none=True)
        out = conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=x, p_neighborhood=nbh)
        out.backward(g)
        return [t.detach().clone() for t in (out, x.grad, conv.proj_axes_.grad, conv.proj_biases_.grad, conv.conv_weights_.grad)]

    want = run(ref_nbh)
    nbh = amd.pc.BQNeighborhood(pc_in, pc_out, r, p_capacity=int(e * 1.3) + 11)
    assert not nbh.symmetric_ and not nbh.overflowed() and nbh.num_edges() == e
    if poison:  # what an uninitialised allocation may hold: ids that look like real points
        tail = nbh.neighbors_i32_[e:]
        tail[:, 0] = torch.randint(0, n_out, (tail.shape[0],), device=DEV, dtype=torch.int32)
        tail[:, 1] = torch.randint(0, n_in, (tail.shape[0],), device=DEV, dtype=torch.int32)
    got = run(nbh)
    for a, b, name in zip(got, want, ("out", "dX", "dA", "dbeta", "dW")):
        assert torch.equal(a, b), name
    # against the oracle as well (the two-phase path is not the only witness)
    ref = O.conv_forward_backward(pc_in.pts_.cpu(), pc_out.pts_.cpu(), pc_in.local_frames_.cpu(), pc_out.local_frames_.cpu(),
                                  ref_nbh.neighbors_.cpu(), x.detach().cpu(), conv.proj_axes_.detach().cpu(),
                                  conv.proj_biases_.detach().cpu(), conv.conv_weights_.detach().cpu(), 1.0 / r, n_out / e, g.cpu())
    for a, b, name in zip(got, ref, ("out", "dX", "dA", "dbeta", "dW")):
        assert rel_err(a, b) < 5e-5, name


@pytest.mark.parametrize("batches", [1, 2, 5])
def test_queries_that_share_a_source_grid_find_the_same_lists(amd, batches):
    """se3_ball_query_bounded_shared: the second and third query against one source cloud with one radius reuse the grid the
    first one sorted -- same lists, bit for bit, as three independent queries; another radius, a changed cloud and a
    capture get a grid of their own."""
    from se3conv3d_amd import pc as PC

    g = torch.Generator().manual_seed(40 + batches)
    n = 9000
    pts = torch.rand(n, 3, generator=g)
    bid = torch.sort(torch.randint(0, batches, (n,), generator=g, dtype=torch.int32)).values
    bid[-1] = batches - 1
    src = PC.Pointcloud(pts.to(DEV), bid.to(DEV))
    dsts = []
    for m in (9000, 2500, 700):
        pd = torch.rand(m, 3, generator=g)
        bd = torch.sort(torch.randint(0, batches, (m,), generator=g, dtype=torch.int32)).values
        dsts.append(PC.Pointcloud(pd.to(DEV), bd.to(DEV)))
    r = 0.06
    box = src.aabb()
    holder = amd.ops.source_grids(src)
    assert holder is amd.ops.source_grids(src) and holder.grids == {}
    for i, dst in enumerate([src] + dsts):
        args = (src.pts_, dst.pts_, src.batch_ids_, dst.batch_ids_, r)
        nb0, ends0, info0 = amd.ops.ball_query_bounded(*args, capacity=400000, n_batches=batches, src_box=box)
        res = amd.ops.ball_query_bounded(*args, capacity=400000, n_batches=batches, src_box=box, grids=holder,
                                         want_sources=dst is src)
        e = int(info0[0])
        assert e > 0 and info0.tolist() == res[2].tolist()
        assert torch.equal(res[0][:e], nb0[:e]) and torch.equal(res[1], ends0)
        if dst is src:
            assert torch.equal(res[3][:e], nb0[:e, 1])
        assert len(holder.grids) == 1
    buf = holder.grids[r][1]
    # the neighbourhood classes pass the holder themselves; another radius is another grid
    nbh = PC.BQNeighborhood(src, dsts[0], r, p_capacity=400000)
    ref, ends_ref = amd.ops.ball_query(src.pts_, dsts[0].pts_, src.batch_ids_, dsts[0].batch_ids_, r, batches)
    assert torch.equal(nbh.neighbors_i32_[:ref.shape[0]], ref) and torch.equal(nbh.start_ids_, ends_ref)
    assert holder.grids[r][1] is buf
    PC.BQNeighborhood(src, dsts[1], 2 * r, p_capacity=900000)
    assert set(holder.grids) == {r, 2 * r}
    # points changed in place: the version counter invalidates the grid
    src.pts_.mul_(0.5)
    nbh2 = PC.BQNeighborhood(src, dsts[0], r, p_capacity=400000)
    ref2, _ = amd.ops.ball_query(src.pts_, dsts[0].pts_, src.batch_ids_, dsts[0].batch_ids_, r, batches)
    assert torch.equal(nbh2.neighbors_i32_[:ref2.shape[0]], ref2) and holder.grids[r][1] is not buf
