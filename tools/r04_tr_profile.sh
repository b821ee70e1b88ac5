#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4y; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_down_up.py tests/test_gpu_bounded_query.py -m gpu -q -x > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"; grep -m3 "Error\|assert" $out/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o du -- python3 tools/profile_down_up.py --reps 10 > $out/du.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/down_up_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$out/down_up_kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('tr_','wrapped_scan','init_lookback')):
        print(f"{n[:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
