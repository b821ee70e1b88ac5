#!/usr/bin/env python3
"""Per-launch HBM traffic of the hot kernels from the rocprofv3 --pmc passes of tools/pmc_passes.sh
(FETCH_SIZE / WRITE_SIZE, KiB per dispatch summed over XCDs) -> profiles/<name>.json, keyed by the stage tags
bench.py reports.  hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE: the x2 is the gfx950 correction of
MI355X_MICROARCH.md (verified on the 16-byte-per-lane streams of this library; uncalibrated for the 4/8-byte
gathers of the edge kernels, whose true read traffic lies between fetch_raw and 2 * fetch_raw).
usage: tools/pmc_traffic.py gpurun_out/pmc_<tag> profiles/r01_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGES = {
    "edge_t_pair_bf16_kernel": ["edge_t_fwd", "edge_t_transposed"],
    "edge_param_grad_bf16_v2_kernel": ["edge_param_grad"],
    "gemm_nn_bf16_kernel": ["gemm_out", "gemm_gradX"],
    "gemm_nn_t24_kernel": ["gemm_out", "gemm_gradX"],
    "gemm_strip_bf16_kernel": ["gemm_gradT"],
    "gemm_tn_bf16_kernel": ["gemm_gradW"],
    "prep_batch_kernel": ["prep"],
}


def main(d, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                for key in STAGES:
                    if key in r["Kernel_Name"]:
                        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # library_source_sha: bench.py reports this file's numbers only while the kernel sources still match

    res = {"_note": __doc__.split("usage:")[0].strip() + "  Headline shape N=65536, k=32, F=2, C=64, K=32, bf16x3.",
           "_library_source_sha": bench.library_source_sha()}
    for key, cs in acc.items():
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024
        write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
        for tag in STAGES[key]:
            res[tag] = {"kernel": key, "fetch_raw": int(fetch), "write": int(write), "hbm_bytes": int(2 * fetch + write)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
