#!/bin/bash
# copies the results of tools/r04_close.sh (merged back under gpurun_out/) into profiles/
set -eu
P=gpurun_out/prof_r04f; C=gpurun_out/r4close
cp $C/bench.json profiles/r04_bench.json
cp $P/stats_kernel_stats.csv profiles/r04_kernel_stats.csv
cp $P/layer_kernel_stats.csv profiles/r04_layer_kernel_stats.csv
cp $P/pmc_summary.txt profiles/r04_pmc_summary.txt
cp $P/traffic.json profiles/r04_traffic.json
cp $P/errors.txt profiles/r04_errors_vs_fp64_oracle.txt
cp gpurun_out/r4close_workloads/summary.txt profiles/r04_workloads.txt
cp gpurun_out/r4close_levels/levels.txt profiles/r04_levels.txt
cp $C/down_up_headline_kernel_stats.csv profiles/r04_down_up_kernel_stats.csv
cp $C/down_up_dfaust_f2_kernel_stats.csv profiles/r04_down_up_dfaust_f2_kernel_stats.csv
(echo "# random-shape parity sweep of the final round-4 code, three arithmetic modes: tools/fuzz_parity.py 48 20261006"; cat $C/fuzz.txt; echo; echo "# the same with SE3_DX_PATH=1 (feature gradient edge-major wherever it is implemented): tools/fuzz_parity.py 48 20261007"; cat $C/fuzz_dx.txt) > profiles/r04_fuzz_parity.txt
python3 - <<'PY'
import json, hashlib, os
r = json.loads(open('profiles/r04_bench.json').read().strip().splitlines()[-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'layer', r['single_layer']['ms_per_step'], 'traffic_source', r['roofline']['traffic_source'][:90])
PY
