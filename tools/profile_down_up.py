#!/usr/bin/env python3
"""The two level-to-level convolutions of a workload's hierarchy (level 0 -> 1 down, level 1 -> 0 up), forward + backward
with the ball query and the source-major transposition inside every repetition: the target program for rocprofv3
--kernel-trace --stats (profiles/r04_down_up_kernel_stats.csv, see profiles/README.md)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--workload", default="headline")
    args = ap.parse_args()
    import se3conv3d_amd as amd
    from se3conv3d_amd import workloads as W

    amd.set_precision(args.precision)
    recs = W.build_down_up(W.WORKLOADS[args.workload], torch.device("cuda", 0), seed=0)
    for _ in range(args.reps):
        for rec in recs:
            nb = amd.pc.BQNeighborhood(rec["pc_in"], rec["pc_out"], rec["r"], p_capacity=int(rec["e"] * 1.25) + 64)
            bench.step_two_clouds(rec, nb)
    torch.cuda.synchronize()
    print("done", [(r["name"], r["n_in"], r["n_out"], r["e"]) for r in recs])


if __name__ == "__main__":
    main()
