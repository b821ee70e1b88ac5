"""CPU: the oracle on the 21 convolution calls of the reference's own FAUST network (BASELINE.json config 2).

tests/golden/network_faust_calls.npz = what the reference's FPNSegUNetMLPGeluRotEqFAUST (models/FPNSegUNet.py:198-223,
Encoder.py:116-173, FPNDecoder.py:87-137, tasks/SemSeg/seg_models.py:16-108) fed to and got from each of its
PNEConvLayerRotEquiv calls in one training-mode forward + backward (tools/gen_golden.py `network_case`): C_in = 1 -> 32,
32 <-> 64 <-> 128 <-> 256 down / up / lateral convolutions, levels of 493 down to 3 points, two bodies, F = 2, PCA frames."""
import pytest
import torch

from conftest import check_weight_gradient, network_calls, rel_err
from oracle import se3conv_oracle as O

TOL = 2e-6


@pytest.fixture(scope="module")
def calls():
    return network_calls()


def test_fixture_holds_the_networks_21_calls(calls):
    assert len(calls) == 21
    shapes = {(c["c_in"], c["c_out"]) for c in calls}
    assert {(1, 32), (32, 64), (64, 128), (128, 256), (256, 256), (256, 128), (128, 64), (64, 32), (32, 32)} <= shapes
    assert sum(not c["same_cloud"] for c in calls) >= 9  # down / up / lateral / output convolutions


@pytest.mark.parametrize("i", range(21))
def test_oracle_reproduces_call(calls, i):
    d = calls[i]
    out, dx, da, db, dw = O.conv_forward_backward(
        d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(), d["x"],
        d["proj_axes"], d["proj_biases"], d["conv_weights"], d["rho"], d["nu"], d["grad_out"])
    assert out.shape == d["out"].shape
    assert rel_err(out, d["out"]) < TOL
    if "dx" in d:
        assert rel_err(dx, d["dx"]) < TOL
    assert rel_err(da, d["dA"]) < TOL
    assert rel_err(db, d["dbeta"]) < TOL
    check_weight_gradient(dw, d, TOL)
