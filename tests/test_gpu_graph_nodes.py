"""GPU: what a captured step of the library is made of -- no memset nodes.

Round 4 found that a HIP graph with MEMSET nodes faults on replay once an RCCL collective has run between two replays (on
the HIP runtime PyTorch 2.10 ships): the library issues no hipMemsetAsync and keeps every sort on
rocPRIM's merge-sort forms.  tests/test_abi.py checks the sources for the call; this file checks the GRAPHS: the step is
captured with `torch.cuda.CUDAGraph(keep_graph=True)`, the raw hipGraph_t is walked with hipGraphGetNodes /
hipGraphNodeGetType, and no node may be of type memset -- for the capacity-bounded ball query (sub-limit sort, the guard of
sort_pairs_no_scratch: ADVICE r4), the transposition between two clouds, the hierarchy step and a whole forward + backward.
The same walk names the torch ops that DO lower to memset nodes inside a capture (INTEGRATION.md lists them).
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HIP_GRAPH_NODE_TYPE_MEMSET = 2  # hipGraphNodeTypeMemset (hip_runtime_api.h: Kernel 0, Memcpy 1, Memset 2, Host 3, ...)


def node_types(graph: torch.cuda.CUDAGraph):
    hip = C.CDLL("libamdhip64.so")  # the runtime torch has already loaded
    hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    hip.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    raw = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, C.byref(n)) == 0
    out = []
    for i in range(n.value):
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t)) == 0
        out.append(t.value)
    return out


def capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()  # eager warm-up: allocator, lazily built lists, the library's spare side streams
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    return g


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd

    se3conv3d_amd.set_precision("bf16x3")
    return se3conv3d_amd


def test_the_walk_sees_memset_nodes(amd):
    """Control: a hipMemsetAsync issued inside a capture IS found as a memset node by this walk; and which torch ops
    produce such nodes on this runtime is printed (INTEGRATION.md quotes the list; -s shows it)."""
    x = torch.empty(1 << 20, device=DEV)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]

    def raw_memset():
        x.add_(1.0)  # a kernel node beside it
        assert hip.hipMemsetAsync(C.c_void_p(x.data_ptr()), 0, x.numel() * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0

    types = node_types(capture(raw_memset))
    assert types.count(HIP_GRAPH_NODE_TYPE_MEMSET) == 1 and len(types) >= 2, types
    found = {}
    for name, fn in (("tensor.zero_()", lambda: x.zero_()), ("torch.zeros(n)", lambda: torch.zeros(1 << 16, device=DEV)),
                     ("tensor.fill_(0.0)", lambda: x.fill_(0.0)), ("tensor.fill_(1.0)", lambda: x.fill_(1.0)),
                     ("torch.zeros_like(x)", lambda: torch.zeros_like(x)), ("x[:1024].zero_()", lambda: x[:1024].zero_()),
                     ("torch.empty(n)", lambda: torch.empty(1 << 16, device=DEV))):
        types = node_types(capture(fn))
        found[name] = (types.count(HIP_GRAPH_NODE_TYPE_MEMSET), len(types))
    print("(memset nodes, all nodes) per captured torch op:", found)
    assert found["torch.empty(n)"][0] == 0


def test_bounded_ball_query_and_transposition_capture_without_memset(amd):
    g = torch.Generator().manual_seed(0)
    n = 20000  # below rocPRIM's merge-sort limit: the sort goes through the radix entry point's merge-sort form
    pts = torch.rand(n, 3, generator=g).to(DEV)
    bid = torch.zeros(n, dtype=torch.int32, device=DEV)
    pc = amd.pc.PointcloudRotEquiv(pts, bid, {"pca": False, "n_frames": 2, "fixed_axis": False})
    sub = amd.pc.PointcloudRotEquiv(pts[: n // 4].contiguous(), bid[: n // 4].contiguous(), {"pca": False, "n_frames": 2, "fixed_axis": False})
    from se3conv3d_amd.workloads import radius_for_degree

    r = radius_for_degree(n, 16)
    held = []

    def step():
        nb = amd.pc.BQNeighborhood(pc, pc, r, p_capacity=n * 24)
        nb2 = amd.pc.BQNeighborhood(pc, sub, r, p_capacity=n * 8)  # two clouds: backward reads the transposed list
        geom = amd.layers._geometry_of(pc, sub, nb2)
        held[:] = [nb, nb2, geom.transpose()]

    types = node_types(capture(step))
    assert len(types) > 5 and types.count(HIP_GRAPH_NODE_TYPE_MEMSET) == 0, types
    assert int(held[0].edge_info_[1]) == 0 and int(held[1].edge_info_[1]) == 0


def test_forward_backward_captures_without_memset(amd):
    from se3conv3d_amd import workloads as W

    spec = dict(W.WORKLOADS["headline"])
    spec["points"] = 8192
    levels = W.build_stack(spec, torch.device(DEV), seed=0)

    def step():
        for lv in levels:
            lv["x"].grad = None
            for p in lv["conv"].parameters():
                p.grad = None
            out = lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])
            out.backward(lv["g"])

    types = node_types(capture(step))
    assert len(types) > 20 and types.count(HIP_GRAPH_NODE_TYPE_MEMSET) == 0, types
