#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4z; mkdir -p $out
SE3_TR_MERGE_SORT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o du -- python3 tools/profile_down_up.py --reps 10 > $out/du.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/down_up_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$out/down_up_kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('split_edges','group_ends','merge_sort','block_sort','Memcpy','copyBuffer')):
        print(f"{n[:140]:140s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
