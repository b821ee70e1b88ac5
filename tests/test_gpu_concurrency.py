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
    """The two-stream backward pass is opt-in since round 5 (it lost its A/B at every size): this module, which tests the
    side-stream machinery, turns it on for layers of up to 32 k output rows and restores the default afterwards."""
    import se3conv3d_amd as amd
    from se3conv3d_amd import _lib

    amd.set_precision("bf16x3")
    _lib.load().se3_set_overlap_rows(32768)
    yield amd
    _lib.load().se3_set_overlap_rows(-1)


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


def _side_stats():
    import ctypes as C
    from se3conv3d_amd import _lib

    buf = (C.c_int32 * 5)()
    _lib.check(_lib.load().se3_side_stream_stats(C.cast(buf, C.c_void_p)), "se3_side_stream_stats")
    return list(buf)[:3]   # [streams that own a side stream, spares on this device, sets created in this process]


def _side_stats_all():
    import ctypes as C
    from se3conv3d_amd import _lib

    buf = (C.c_int32 * 5)()
    _lib.check(_lib.load().se3_side_stream_stats(C.cast(buf, C.c_void_p)), "se3_side_stream_stats")
    return list(buf)   # ... + [forks skipped inside a capture, sets taken back by the 16-owner cap]


def test_capture_creates_no_runtime_objects(amd):
    """The capture contract (include/se3conv.h): the side stream a captured backward forks onto was made by
    an earlier EAGER call -- nothing is created while the caller's stream is being captured."""
    c = _case(amd, 6)                       # 10 000 output rows: backward forks
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):              # warm-up on a side stream, as torch asks for before a capture
        ref = _fwd_bwd(c)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    owners0, spares0, created0 = _side_stats()
    assert spares0 >= 1
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):           # torch captures on a stream of its own: new to the library
        outs = _fwd_bwd(c)
    owners1, spares1, created1 = _side_stats()
    assert created1 == created0, "a stream / event was created during the capture"
    # torch captures on a stream from its pool: new to the library (it took a spare) or known from an earlier capture
    assert (owners1, spares1) in ((owners0 + 1, spares0 - 1), (owners0, spares0))
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for u, v in zip(outs, ref):
        assert torch.equal(u, v)
    _fwd_bwd(c)                             # the next eager call tops the spares up again
    assert _side_stats()[1] >= spares0


def test_first_call_of_a_process_inside_a_capture_does_not_fork():
    """No eager call before the capture: no spare exists, the captured backward runs its branches back to back (same
    results).  Own process: the library state of this one is warm."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import se3conv3d_amd as amd
import test_gpu_concurrency as T
amd.set_precision("bf16x3")
from se3conv3d_amd import _lib
_lib.load().se3_set_overlap_rows(32768)   # the two-stream backward pass is opt-in
# geometry, neighbourhood and parameters are built eagerly (they are inputs); the operator itself first runs captured
c = T._case(amd, 7)
assert T._side_stats() == [0, 0, 0]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    outs = T._fwd_bwd(c)
assert T._side_stats() == [0, 0, 0], T._side_stats()
assert T._side_stats_all()[3] == 1, T._side_stats_all()   # the library says that this graph runs its branches back to back
graph.replay(); torch.cuda.synchronize()
ref = T._fwd_bwd(c); torch.cuda.synchronize()
assert all(torch.equal(u, v) for u, v in zip(outs, ref))
assert T._side_stats()[2] > 0
print("ok")
''' % (root, root)
    proc = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "ok" in proc.stdout, (proc.stdout + proc.stderr)[-2000:]


def test_side_stream_table_is_capped(amd):
    """A process that makes a stream per request: at most 16 caller streams own a side stream, the least recently used
    sets go back to the spares and are handed out again -- the table and the number of runtime objects stop growing."""
    c = _case(amd, 8)                       # 10 000 output rows: backward forks
    ref = _fwd_bwd(c)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(24)]
    created = []
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs = _fwd_bwd(c)
        s.synchronize()
        for u, v in zip(outs, ref):
            assert torch.equal(u, v)
        created.append(_side_stats_all()[2])
    owners, spares, made, _, evicted = _side_stats_all()
    assert owners <= 16 + 1, (owners, spares, made, evicted)    # + 1: torch's capture stream of an earlier test may still own one
    assert evicted >= 24 - 17
    assert created[-1] == created[-4], "runtime objects are still being created although the cap hands sets back"


def test_tensors_on_another_device_are_refused(amd):
    from se3conv3d_amd import ops

    with pytest.raises(ValueError, match="current device"):
        ops._stream(torch.device("cuda", torch.cuda.current_device() + 1))


def test_more_concurrent_backward_calls_than_the_cap(amd):
    """with more caller streams in flight than the 16-owner cap, the eviction must never hand a side-stream set
    that is between its fork and its join to another caller (both calls would record and wait on the same events, and one
    call's side-branch kernels could start before its own preparation had finished: silently wrong gradients).  20 threads,
    each on a stream of its own, run backward at the same time, several rounds; every result equals the serial one bit for bit."""
    import threading

    n_threads = 20
    cases = [_case(amd, 20 + (i % 4), n=5000) for i in range(4)]   # 10 000 output rows: backward forks
    refs = [_fwd_bwd(c) for c in cases]
    torch.cuda.synchronize()
    # one set of tensors per thread (shared clouds / neighbourhoods / parameters are read-only; grads are per thread)
    work = []
    for i in range(n_threads):
        c = cases[i % 4]
        conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(64, 64).to(DEV)
        conv.load_state_dict(c["conv"].state_dict())
        work.append(dict(pc=c["pc"], nbh=c["nbh"], conv=conv, x=c["x"].detach().clone().requires_grad_(True), g=c["g"]))
    got, errs = [None] * n_threads, []
    start = threading.Barrier(n_threads)

    def run(i):
        try:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.default_stream())
            start.wait()
            with torch.cuda.stream(st):
                for _ in range(6):
                    got[i] = _fwd_bwd(work[i])
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(n_threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(n_threads):
        for u, v in zip(got[i], refs[i % 4]):
            assert torch.equal(u, v), f"thread {i}"
    assert _side_stats_all()[0] <= 16 + n_threads  # owners never shrink below what is pinned; nothing leaked beyond the threads
