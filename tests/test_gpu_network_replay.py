"""GPU: replay of the reference network's own convolution calls (BASELINE.json config 2) and of drop-in path A.

(1) `network_faust_calls.npz` (tools/gen_golden.py `network_case`): what the reference's FPNSegUNetMLPGeluRotEqFAUST
    (models/FPNSegUNet.py:198-223, Encoder.py:116-173, FPNDecoder.py:87-137, tasks/SemSeg/seg_models.py:72-101) fed to and
    got from each of its 21 PNEConvLayerRotEquiv calls -- C_in = 1 -> 32, 32 <-> 64 <-> 128 <-> 256 down / up / lateral
    convolutions on levels of 493 ... 3 points -- replayed through `amd.PNEConvLayerRotEquivFactory` on the HIP path.
    The layer gets FOREIGN objects, as when the reference's own containers are handed over (path A):
    `types.SimpleNamespace` clouds carrying `pts_`, `local_frames_`, `n_frames_` only, a neighbourhood carrying only the
    reference's int64 `neighbors_`, `start_ids_`, `radius_`.  Every unique neighbourhood is also rebuilt by
    `amd.pc.BQNeighborhood` and compared as an edge set.
(2) three layer fixtures (same cloud, down-convolution, sparse rows) through the same foreign objects with the rows of
    every sample SHUFFLED -- the reference's store pass leaves them in the order its atomics were served
    (custom_ops/ball_query/store_neighbors.cu:129-175).
"""
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import GOLDEN, NETWORK_FIXTURES, canon_edges, check_recorded, check_weight_gradient, load_npz, network_calls, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOLS = {"fp32": 2e-5, "bf16x3": 5e-5}


@pytest.fixture(scope="module", params=["bf16x3", "fp32"])
def amd(built_library, request):
    import se3conv3d_amd

    se3conv3d_amd.set_precision(request.param)
    yield se3conv3d_amd
    se3conv3d_amd.set_precision("bf16x3")


_calls = {}


def calls_of(net):
    if net not in _calls:
        _calls[net] = network_calls(net)
    return _calls[net]


def foreign_cloud(pts, frames):
    return SimpleNamespace(pts_=pts.to(DEV), local_frames_=frames.to(DEV), n_frames_=int(frames.shape[1]))


def foreign_neighbourhood(neighbors, ends, radius):
    return SimpleNamespace(neighbors_=neighbors.to(torch.int64).to(DEV), start_ids_=ends.to(DEV), radius_=radius)


def layer_of(amd, d):
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(d["c_in"] if "c_in" in d else d["conv_weights"].shape[0],
                                                                                d["conv_weights"].shape[2])
    conv.load_state_dict({"proj_axes_": d["proj_axes"], "proj_biases_": d["proj_biases"], "conv_weights_": d["conv_weights"],
                          "norm_neigh_dist_": d["rho"], "norm_num_neighs_": d["nu"]})
    return conv.to(DEV)


def run_and_check(amd, d, pc_in, pc_out, nbh, check_dw):
    tol = TOLS[amd.get_precision()]
    conv = layer_of(amd, d)
    want_dx = "dx" in d or "dx_at" in d
    x = d["x"].to(DEV).requires_grad_(want_dx)
    out = conv(p_pc_in=pc_in, p_pc_out=pc_out, p_in_features=x, p_neighborhood=nbh)
    assert out.dtype == torch.float32 and out.is_cuda
    out.backward(d["grad_out"].to(DEV))
    check_recorded(out, d, "out", tol)
    if want_dx:
        check_recorded(x.grad, d, "dx", tol)
    assert rel_err(conv.proj_axes_.grad, d["dA"]) < tol
    assert rel_err(conv.proj_biases_.grad, d["dbeta"]) < tol
    check_dw(conv.conv_weights_.grad, tol)


@pytest.mark.parametrize("net,i", [(net, i) for net, (_, _, n) in NETWORK_FIXTURES.items() for i in range(n)])
def test_reference_network_call_replayed_on_the_hip_path(amd, net, i):
    """FAUST: 21 calls (config 2).  ScanNet (round 6): the 32 calls of FPNSegUNetMLPGeluRotEqScanNet (config 3) -- F = 1 frames
    about the up axis, widths 64 ... 320, FPN laterals from every level to the finest one."""
    d = calls_of(net)[i]
    pc_in = foreign_cloud(d["pts_in"], d["frames_in"])
    pc_out = pc_in if d["same_cloud"] else foreign_cloud(d["pts_out"], d["frames_out"])
    nbh = foreign_neighbourhood(d["neighbors"], d["ends"], d["radius"])
    run_and_check(amd, d, pc_in, pc_out, nbh, lambda got, tol: check_weight_gradient(got, d, tol))


@pytest.mark.parametrize("net,n_unique", [("faust", 12), ("scannet", 17)])
def test_network_neighbourhoods_rebuilt_by_the_library(amd, net, n_unique):
    """Every neighbourhood the reference networks built (ball queries between levels of 493 ... 3 points, two bodies; 490 ... 2
    points, two rooms): the library's own query gives the same edge set and the same offsets."""
    if amd.get_precision() != "bf16x3":
        pytest.skip("integer work: one arithmetic mode is enough")
    seen = set()
    for d in calls_of(net):
        key = (d["neighbors"].data_ptr(),)
        if key in seen:
            continue
        seen.add(key)
        pc_in = amd.pc.PointcloudRotEquiv.from_frames(d["pts_in"].to(DEV), d["batch_in"].to(DEV), d["frames_in"].to(DEV))
        pc_out = pc_in if d["same_cloud"] else amd.pc.PointcloudRotEquiv.from_frames(
            d["pts_out"].to(DEV), d["batch_out"].to(DEV), d["frames_out"].to(DEV))
        nbh = amd.pc.BQNeighborhood(pc_in, pc_out, d["radius"])
        assert torch.equal(nbh.start_ids_.cpu(), d["ends"])
        assert torch.equal(canon_edges(nbh.neighbors_), canon_edges(d["neighbors"]))
    assert len(seen) >= n_unique


SHUFFLED = ["layer_n256_f2_c64.npz", "layer_down_n512_n128_f2.npz", "layer_sparse_n200_f2.npz"]


@pytest.mark.parametrize("name", SHUFFLED)
def test_foreign_objects_with_rows_shuffled_inside_each_sample(amd, name):
    """Drop-in path A: nothing but the reference's attributes (no `neighbors_i32_`, `edge_info_`, `symmetric_`,
    `sources_i32_`, `source_major`), int64 edges in arbitrary order inside a sample."""
    d = load_npz(os.path.join(GOLDEN, name))
    nb, ends = d["neighbors"].clone(), d["ends"]
    g = torch.Generator().manual_seed(5)
    start = 0
    for e in ends.tolist():
        if e - start > 1:
            nb[start:e] = nb[start:e][torch.randperm(e - start, generator=g)]
        start = e
    assert not torch.equal(nb, d["neighbors"]) and torch.equal(canon_edges(nb), canon_edges(d["neighbors"]))
    same = d["pts_in"].shape == d["pts_out"].shape and torch.equal(d["pts_in"], d["pts_out"]) \
        and torch.equal(d["frames_in"], d["frames_out"])
    pc_in = foreign_cloud(d["pts_in"], d["frames_in"])
    pc_out = pc_in if same else foreign_cloud(d["pts_out"], d["frames_out"])
    nbh = foreign_neighbourhood(nb, ends, float(d["radius"]))
    assert sorted(vars(nbh)) == ["neighbors_", "radius_", "start_ids_"] and sorted(vars(pc_in)) == ["local_frames_", "n_frames_", "pts_"]
    run_and_check(amd, d, pc_in, pc_out, nbh, lambda got, tol: rel_err(got, d["dW"]) < tol or pytest.fail("dW"))
