#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_<tag>/*_counter_collection.csv): per kernel, the mean
of every counter over its dispatches (+ mean duration from the kernel trace of the same pass)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?(?:se3::)?([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:44]


def main(d):
    table = defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        trace = f.replace("_counter_collection", "_kernel_trace")
        dur = defaultdict(list)
        if os.path.exists(trace):
            for r in csv.DictReader(open(trace)):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, cs in acc.items():
            for c, v in cs.items():
                table[k][c] = sum(v) / len(v)
            if k in dur:
                table[k]["us(" + os.path.basename(f).split("_")[0] + ")"] = sum(dur[k]) / len(dur[k])
    keep = [k for k in table if any(s in k for s in ("edge", "gemm", "split", "scan_cand", "find_ranges"))]
    for k in sorted(keep):
        print(k)
        for c, v in sorted(table[k].items()):
            print(f"    {c:28s} {v:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
