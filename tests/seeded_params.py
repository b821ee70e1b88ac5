"""Parameters of the network fixtures' convolutions, re-drawn from seeds (shared by the generator tools/gen_golden.py, which
runs in the build container only, and by the tests that replay the fixtures: the 21 / 32 convolutions of the reference's FAUST /
ScanNet networks hold 9.2 M / 40 M weights, the fixtures store their sums)."""
import numpy as np
import torch


def seeded_conv_params(index, dims, c_in, num_basis, c_out, base=7000):
    """Same distributions as the reference's init (PNEConvLayer.py:79-88), biases moved off zero."""
    g = torch.Generator().manual_seed(base + index)
    ba, bw = float(np.sqrt(1.0 / dims)), float(np.sqrt(1.0 / (c_in * num_basis)))
    axes = (torch.rand(dims, num_basis, generator=g) * 2 - 1) * ba
    biases = (torch.rand(num_basis, generator=g) * 2 - 1) * 0.5
    weights = (torch.rand(c_in, num_basis, c_out, generator=g) * 2 - 1) * bw
    return axes, biases, weights
