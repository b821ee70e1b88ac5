"""CPU: the oracle on the convolution calls of the reference's own networks (BASELINE.json configs 2 and 3).

tests/golden/network_faust_calls.npz = what the reference's FPNSegUNetMLPGeluRotEqFAUST (models/FPNSegUNet.py:198-223,
Encoder.py:116-173, FPNDecoder.py:87-137, tasks/SemSeg/seg_models.py:16-108) fed to and got from each of its 21
PNEConvLayerRotEquiv calls in one training-mode forward + backward (tools/gen_golden.py `network_case`): C_in = 1 -> 32,
32 <-> 64 <-> 128 <-> 256 down / up / lateral convolutions, levels of 493 down to 3 points, two bodies, F = 2, PCA frames.

tests/golden/network_scannet_calls.npz (round 6) = the same for FPNSegUNetMLPGeluRotEqScanNet (seg_models.py:39-58,90-95;
confs/scannet/scannet20_rot_pca_SO2.yaml:26-41: blocks [2,3,4,6,4], widths [64,128,192,256,320], FPN width 128, F = 1 PCA
frames about the fixed axis 2, grids 0.1 ... 1.6) fed as tasks/SemSeg/train_scannet_rot.py:142-186,262-290 does, on two
synthetic rooms: 32 calls on levels of 490 down to 2 points; outputs / feature gradients above 4 096 entries are kept at
2 048 seeded positions + their norm."""
import pytest
import torch

from conftest import NETWORK_FIXTURES, check_recorded, check_weight_gradient, network_calls, rel_err
from oracle import se3conv_oracle as O

TOL = 2e-6
CASES = [(net, i) for net, (_, _, n) in NETWORK_FIXTURES.items() for i in range(n)]
_calls = {}


def calls_of(net):
    if net not in _calls:
        _calls[net] = network_calls(net)
    return _calls[net]


def test_fixture_holds_the_faust_networks_21_calls():
    calls = calls_of("faust")
    assert len(calls) == 21
    shapes = {(c["c_in"], c["c_out"]) for c in calls}
    assert {(1, 32), (32, 64), (64, 128), (128, 256), (256, 256), (256, 128), (128, 64), (64, 32), (32, 32)} <= shapes
    assert sum(not c["same_cloud"] for c in calls) >= 9  # down / up / lateral / output convolutions


def test_fixture_holds_the_scannet_networks_32_calls():
    calls = calls_of("scannet")
    assert len(calls) == 32
    shapes = {(c["c_in"], c["c_out"]) for c in calls}
    assert {(64, 64), (64, 128), (128, 128), (128, 192), (192, 192), (192, 256), (256, 256), (256, 320), (320, 320), (320, 256),
            (256, 192), (192, 128), (128, 64)} <= shapes
    assert all(c["frames_in"].shape[1] == 1 and c["frames_out"].shape[1] == 1 for c in calls)  # train_n_frames: 1
    assert sum(not c["same_cloud"] for c in calls) >= 13  # 4 down, 4 up, 4 FPN laterals, the output convolution
    # frames about the fixed up axis: third column = +e_z (RotationFunctions.py:383-404)
    fr = calls[0]["frames_in"].reshape(-1, 3, 3)
    assert torch.allclose(fr[:, :, 2], torch.tensor([0.0, 0.0, 1.0]).expand(fr.shape[0], 3), atol=1e-6)


@pytest.mark.parametrize("net,i", CASES)
def test_oracle_reproduces_call(net, i):
    d = calls_of(net)[i]
    out, dx, da, db, dw = O.conv_forward_backward(
        d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(), d["x"],
        d["proj_axes"], d["proj_biases"], d["conv_weights"], d["rho"], d["nu"], d["grad_out"])
    check_recorded(out, d, "out", TOL)
    if "dx" in d or "dx_at" in d:
        check_recorded(dx, d, "dx", TOL)
    assert rel_err(da, d["dA"]) < TOL
    assert rel_err(db, d["dbeta"]) < TOL
    check_weight_gradient(dw, d, TOL)
