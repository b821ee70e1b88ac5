import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_layer_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "layer_*.npz")))


def load_npz(path, device="cpu"):
    out = {}
    with np.load(path) as z:
        for k in z.files:
            v = z[k]
            out[k] = torch.from_numpy(v).to(device) if v.ndim else torch.tensor(v.item())
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a - b||_2 / ||b||_2 (the north-star tolerance is stated on this quantity)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


def canon_edges(neighbors: torch.Tensor) -> torch.Tensor:
    """Edge list sorted by (sample, source): order inside a sample is undefined in the reference."""
    nb = neighbors.detach().cpu().to(torch.int64)
    key = nb[:, 0] * (int(nb[:, 1].max()) + 1 if nb.numel() else 1) + nb[:, 1]
    return nb[torch.argsort(key)]


@pytest.fixture(scope="session")
def built_library():
    """Build (or reuse) the HIP library; tests that need it fail loudly if it cannot be built."""
    from se3conv3d_amd import build

    return build.build(verbose=False)


# ----------------------------------------------------------------------- the reference network's own conv calls
NETWORK_FIXTURES = {  # name -> (file, seed base of its parameters, convolution calls)
    "faust": (os.path.join(GOLDEN, "network_faust_calls.npz"), 7000, 21),
    "scannet": (os.path.join(GOLDEN, "network_scannet_calls.npz"), 7100, 32),
}


def network_calls(net="faust"):
    """The convolution calls of the reference's FPNSegUNetMLPGeluRotEqFAUST (21) / FPNSegUNetMLPGeluRotEqScanNet (32)
    recorded by tools/gen_golden.py (`network_case`): a list of dicts with the clouds, the neighbourhood, buffers, input,
    output and the gradients autograd delivered.  The 9.2 M / 40 M weights are not stored: they are re-drawn from their seeds
    (tests/seeded_params.py, the function the generator used) and checked against the stored sums (bit-for-bit the same torch
    CPU generator calls)."""
    from seeded_params import seeded_conv_params

    path, base, _ = NETWORK_FIXTURES[net]
    z = load_npz(path)
    calls = []
    for i in range(int(z["n_calls"])):
        p = f"c{i:02d}/"
        conv_index, ci, co, ni, c_in, c_out = (int(v) for v in z[p + "meta"])
        axes, biases, weights = seeded_conv_params(conv_index, 9, c_in, 32, c_out, base)
        sums = [float(t.double().sum()) for t in (axes, biases, weights)] + [float(weights.double().abs().sum())]
        assert np.allclose(sums, z[p + "param_sums"].numpy(), rtol=1e-12, atol=1e-12), \
            "the seeded parameters of the network fixture did not re-draw identically (another torch CPU generator?)"
        rec = {"index": i, "c_in": c_in, "c_out": c_out, "proj_axes": axes, "proj_biases": biases, "conv_weights": weights,
               "rho": z[p + "rho"], "nu": z[p + "nu"], "same_cloud": ci == co}
        for side, c in (("in", ci), ("out", co)):
            rec["pts_" + side], rec["batch_" + side] = z[f"cloud{c}/pts"], z[f"cloud{c}/batch"]
            rec["frames_" + side] = z[f"cloud{c}/frames"]
        rec["neighbors"], rec["ends"], rec["radius"] = z[f"nbh{ni}/neighbors"], z[f"nbh{ni}/ends"], float(z[f"nbh{ni}/radius"])
        for k in ("x", "out", "grad_out", "dx", "dA", "dbeta", "dW", "dW_pos", "dW_at", "dW_norm", "out_pos", "out_at", "out_norm",
                  "out_shape", "dx_pos", "dx_at", "dx_norm", "dx_shape"):
            if p + k in z:
                rec[k] = z[p + k]
        calls.append(rec)
    return calls


def check_recorded(got: torch.Tensor, rec, key: str, tol: float):
    """A recorded tensor of a call (`out`, `dx`): in full where the fixture holds it, else at the fixture's sampled positions
    + its norm and shape."""
    if key in rec:
        assert got.shape == rec[key].shape
        assert rel_err(got, rec[key]) < tol
        return
    assert list(got.shape) == rec[key + "_shape"].tolist()
    at = got.detach().reshape(-1).cpu()[rec[key + "_pos"].long()]
    assert rel_err(at, rec[key + "_at"]) < tol
    _same_norm(got, float(rec[key + "_norm"]), tol)


def _same_norm(got: torch.Tensor, want: float, tol: float):
    """(a block the network's drop path removed for every batch element hands its convolution a zero gradient: norm 0)"""
    n = float(got.detach().double().norm())
    assert n == 0.0 if want == 0.0 else abs(n / want - 1.0) < tol


def check_weight_gradient(got: torch.Tensor, rec, tol: float):
    """dW of a recorded call: in full where the fixture holds it, else at the fixture's sampled positions + its norm."""
    if "dW" in rec:
        assert rel_err(got, rec["dW"]) < tol
        return
    at = got.detach().reshape(-1).cpu()[rec["dW_pos"].long()]
    assert rel_err(at, rec["dW_at"]) < tol
    _same_norm(got, float(rec["dW_norm"]), tol)
