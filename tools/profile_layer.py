#!/usr/bin/env python3
"""One full-resolution layer (N=65536, k=32, F=2, C=64) forward+backward, a few repetitions: the
target program for rocprofv3 kernel-trace / --pmc passes (see profiles/README.md)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--workload", default="headline")
    args = ap.parse_args()
    import se3conv3d_amd as amd
    from se3conv3d_amd import workloads as W

    amd.set_precision(args.precision)
    spec = dict(W.WORKLOADS[args.workload])
    if args.workload == "headline":
        spec["points"] = args.n
    levels = W.build_stack(spec, torch.device("cuda", 0), seed=0, n_levels=1)
    for _ in range(args.reps):
        bench.step(levels[:1])
    torch.cuda.synchronize()
    print("done", levels[0]["n"], levels[0]["e"])


if __name__ == "__main__":
    main()
