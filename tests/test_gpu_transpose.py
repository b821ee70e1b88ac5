"""GPU: the source-major copy of an edge list (se3_csr_transpose / se3_csr_transpose_bounded), every form of it against a
stable sort of the list by source: the counting form with its three segment rankings (<= 64 entries by shuffles, <= 2048
through LDS, longer ones from memory), the merge sort (more than two sources per row, and SE3_TR_MERGE_SORT=1 everywhere),
bounded buffers with poisoned tails, the optional edge ids, and the whole thing captured into a HIP graph.  The reference has no such function: its backward scatters with float atomics
(feat_basis_proj_grads.cu:126,140); this list is what replaces them with a deterministic gather."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd as amd
    return amd


def random_list(n_samples, n_src, degree, seed, hub=None):
    """A sample-major edge list: every sample lists `degree` DISTINCT random sources (plus source `hub` when given)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for s in range(n_samples):
        src = torch.randperm(n_src, generator=g)[:degree]
        if hub is not None and hub not in src.tolist():
            src = torch.cat((src, torch.tensor([hub])))
        rows.append(torch.stack((torch.full_like(src, s), src), 1))
    return torch.cat(rows).to(torch.int32)


def reference(nb, n_src):
    order = torch.sort(nb[:, 1].long(), stable=True).indices
    t_samples = nb[order, 0]
    t_ends = torch.cumsum(torch.bincount(nb[:, 1].long(), minlength=n_src), 0).to(torch.int32)
    return t_samples, t_ends


CASES = [
    # n_samples, n_src, degree, hub      which form
    (6000, 300, 20, None),             # few sources, segments of ~400 entries (ranked through LDS): an up-convolution's shape
    (20000, 2000, 4, None),            # segments of ~40 (shuffles)
    (3000, 20000, 12, 7),              # short segments + one of 3000 entries (ranked from memory)
    (4000, 5000, 3, 11),               # segments of 1..64, one of 4000
    (1500, 40000, 2, 5),               # a segment of 1500 (LDS ranking), most sources empty
    (40, 100000, 3, None),             # 120 rows, 100 k sources: the merge-sort form
    (1, 50, 9, None),                  # one sample
]


@pytest.mark.parametrize("n_samples,n_src,degree,hub", CASES)
@pytest.mark.parametrize("bounded", [False, True])
def test_transpose_is_the_stable_sort_by_source(amd, n_samples, n_src, degree, hub, bounded):
    nb = random_list(n_samples, n_src, degree, seed=n_src + degree, hub=hub)
    e = nb.shape[0]
    want_s, want_e = reference(nb, n_src)
    if bounded:  # a capacity-sized buffer whose tail holds ids that look like real points
        g = torch.Generator().manual_seed(3)
        cap = e + e // 3 + 17
        tail = torch.stack((torch.randint(0, n_samples, (cap - e,), generator=g), torch.randint(0, n_src, (cap - e,), generator=g)), 1)
        buf = torch.cat((nb, tail.to(torch.int32))).to(DEV)
        info = torch.tensor([e, 0], dtype=torch.int32, device=DEV)
        ts, te, ti = amd.ops.csr_transpose(buf, n_src, info, want_edge_ids=True)
        assert ts.shape[0] == cap and bool((ts[e:] == 0).all()) and bool((ti[e:] == 0).all()), "rows behind the list must be zeroed"
    else:
        ts, te, ti = amd.ops.csr_transpose(nb.to(DEV), n_src, want_edge_ids=True)
    assert torch.equal(te.cpu(), want_e)
    assert torch.equal(ts[:e].cpu(), want_s)
    # t_edge_ids: entry j of the result is row t_edge_ids[j] of the sample-major list
    ti = ti[:e].cpu().long()
    assert torch.equal(nb[ti, 0], want_s) and torch.equal(ti, torch.sort(nb[:, 1].long(), stable=True).indices)
    # the third result is optional
    ts2, te2 = amd.ops.csr_transpose(nb.to(DEV), n_src)
    assert torch.equal(ts2.cpu(), want_s) and torch.equal(te2.cpu(), want_e)


def test_transpose_of_an_empty_list_and_of_no_sources(amd):
    ts, te = amd.ops.csr_transpose(torch.zeros((0, 2), dtype=torch.int32, device=DEV), 12)
    assert ts.shape[0] == 0 and te.tolist() == [0] * 12
    ts, te = amd.ops.csr_transpose(torch.zeros((0, 2), dtype=torch.int32, device=DEV), 0)
    assert ts.shape[0] == 0 and te.shape[0] == 0


def test_transpose_replays_in_a_graph(amd):
    """No host synchronisation, nothing allocated by the library, and the same answer on every replay (the slots of the
    counting form are handed out by atomics: the ranking behind them must hide their order)."""
    for n_samples, n_src, degree, hub in (CASES[0], CASES[2]):
        nb = random_list(n_samples, n_src, degree, seed=5, hub=hub).to(DEV)
        want_s, want_e = reference(nb.cpu(), n_src)
        held = {}

        def step():
            held["r"] = amd.ops.csr_transpose(nb, n_src)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for _ in range(4):
            held["r"][0].fill_(-7), held["r"][1].fill_(-7)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(held["r"][1].cpu(), want_e) and torch.equal(held["r"][0].cpu(), want_s)


def test_merge_sort_switch_gives_the_same_list(amd):
    """SE3_TR_MERGE_SORT=1 (read once per process: a child) routes every list through the merge-sort form."""
    code = (
        "import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import se3conv3d_amd as amd\n"
        "from test_gpu_transpose import random_list, reference\n"
        "nb = random_list(6000, 300, 20, seed=1)\n"
        "ts, te = amd.ops.csr_transpose(nb.to('cuda:0'), 300)\n"
        "ws, we = reference(nb, 300)\n"
        "assert torch.equal(ts.cpu(), ws) and torch.equal(te.cpu(), we)\n"
        "print('ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, SE3_TR_MERGE_SORT="1"), cwd=ROOT)
    assert p.returncode == 0 and "ok" in p.stdout, (p.stdout[-800:], p.stderr[-800:])
