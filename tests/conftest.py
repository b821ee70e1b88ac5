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
