"""GPU: the C-ABI contract "re-entrant, no shared events or streams between calls" (include/se3conv.h).
se3conv_bwd runs its two branches on an internal side stream for 4 k - 32 k output rows; two backward calls issued
at the same time on two caller streams must not share that stream or its fork / join events: their results equal
the serial ones bit for bit (the kernels are deterministic).  Also: tensors on another device than the current
one are refused instead of being launched on the wrong device's stream."""
import pytest
import torch

from se3conv3d_amd.workloads import radius_for_degree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd as amd
    amd.set_precision("bf16x3")
    return amd


def _case(amd, seed, n=5000, f=2, c=64):
    torch.manual_seed(seed)
    pts = torch.rand(n, 3, device=DEV)
    bid = torch.zeros(n, dtype=torch.int32, device=DEV)
    pc = amd.pc.PointcloudRotEquiv(pts, bid, {"pca": False, "n_frames": f, "fixed_axis": False})
    r = radius_for_degree(n, 24)
    nbh = amd.pc.BQNeighborhood(pc, pc, r)
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c, c).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r)
    conv.norm_num_neighs_.fill_(n / nbh.neighbors_.shape[0])
    x = torch.randn(n * f, c, device=DEV, requires_grad=True)
    g = torch.randn(n * f, c, device=DEV)
    return dict(pc=pc, nbh=nbh, conv=conv, x=x, g=g)


def _fwd_bwd(c):
    c["x"].grad = None
    for p in c["conv"].parameters():
        p.grad = None
    out = c["conv"](p_pc_in=c["pc"], p_pc_out=c["pc"], p_in_features=c["x"], p_neighborhood=c["nbh"])
    out.backward(c["g"])
    return [out.detach().clone(), c["x"].grad.clone()] + [p.grad.clone() for p in c["conv"].parameters()]


def test_two_backward_calls_on_two_streams_equal_the_serial_results(amd):
    a, b = _case(amd, 1), _case(amd, 2)     # 10 000 output rows each: inside the forked range
    ref_a, ref_b = _fwd_bwd(a), _fwd_bwd(b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):                      # interleaved issue: the side work of both calls is in flight together
        s1.wait_stream(torch.cuda.current_stream())
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            got_a = _fwd_bwd(a)
        with torch.cuda.stream(s2):
            got_b = _fwd_bwd(b)
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
        torch.cuda.synchronize()
        for u, v in zip(got_a + got_b, ref_a + ref_b):
            assert torch.equal(u, v)


def test_two_threads_two_streams(amd):
    import threading

    cases = [_case(amd, 3), _case(amd, 4)]
    refs = [_fwd_bwd(c) for c in cases]
    torch.cuda.synchronize()
    got, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(4):
                    got[i] = _fwd_bwd(cases[i])
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        for u, v in zip(got[i], refs[i]):
            assert torch.equal(u, v)


def test_forked_backward_inside_graph_capture(amd):
    c = _case(amd, 5)
    ref = _fwd_bwd(c)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):           # the fork / join must be captured as a joined branch
        _fwd_bwd_outs = _fwd_bwd(c)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for u, v in zip(_fwd_bwd_outs, ref):
        assert torch.equal(u, v)


def test_tensors_on_another_device_are_refused(amd):
    from se3conv3d_amd import ops

    with pytest.raises(ValueError, match="current device"):
        ops._stream(torch.device("cuda", torch.cuda.current_device() + 1))
