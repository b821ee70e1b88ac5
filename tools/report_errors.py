#!/usr/bin/env python3
"""Relative errors (||d||/||ref||) of both arithmetic modes against the fp64 evaluation of the oracle,
on the golden fixtures and a denser random case.  GPU box only; output goes into DESIGN.md."""
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd  # noqa: E402
from oracle import se3conv_oracle as O  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())


def run(d, mode):
    amd.set_precision(mode)
    geom = amd.ops.ConvGeometry.build(d["pts_in"].to(DEV), d["pts_out"].to(DEV), d["frames_in"].to(DEV),
                                      d["frames_out"].to(DEV), d["neighbors"].to(DEV), d["ends"].to(DEV))
    x = d["x"].to(DEV).requires_grad_(True)
    a, b, w = (d[k].to(DEV).requires_grad_(True) for k in ("proj_axes", "proj_biases", "conv_weights"))
    out = amd.SE3ConvFunction.apply(x, a, b, w, geom, d["rho"], d["nu"])
    out.backward(d["grad_out"].to(DEV))
    return out, x.grad, a.grad, b.grad, w.grad


def main():
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "layer_*.npz")))
    print(f"{'case':28s} {'mode':7s} " + " ".join(f"{n:>9s}" for n in ("out", "dX", "dA", "dbeta", "dW")))
    for f in files:
        d = {k: torch.from_numpy(v) if v.ndim else torch.tensor(v.item()) for k, v in np.load(f).items()}
        ref = O.conv_forward_backward(d["pts_in"], d["pts_out"], d["frames_in"], d["frames_out"], d["neighbors"].long(),
                                      d["x"], d["proj_axes"], d["proj_biases"], d["conv_weights"], d["rho"], d["nu"],
                                      d["grad_out"], dtype=torch.float64)
        for mode in ("fp32", "bf16x3", "bf16x3_t16"):
            got = run(d, mode)
            print(f"{os.path.basename(f)[6:-4]:28s} {mode:7s} " + " ".join(f"{rel(u, v):9.2e}" for u, v in zip(got, ref)))
        gold = (d["out"], d["dx"], d["dA"], d["dbeta"], d["dW"])
        print(f"{'  (reference fp32 python)':28s} {'':7s} " + " ".join(f"{rel(u, v):9.2e}" for u, v in zip(gold, ref)))


if __name__ == "__main__":
    main()
