#!/bin/bash
# rocprofv3 kernel statistics of single levels (eager launches):  tools/trace_small_levels.sh <tag> "<workload>:<level> ..."
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
for wl_lv in $*; do
  wl=${wl_lv%%:*}; lv=${wl_lv##*:}
  SE3_OVERLAP_ROWS=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ${wl}_l${lv} -- python3 tools/profile_small_levels.py $wl $lv 20 > $out/${wl}_l${lv}.log 2>&1 || echo "rc=$? $wl_lv"
  f=$(find $out -name "${wl}_l${lv}_kernel_stats.csv" | head -1)
  echo "== $wl level $lv"; tail -1 $out/${wl}_l${lv}.log
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "se3" in r["Name"] or "hipcub" in r["Name"] or "rocprim" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f"  {float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls']):4d}  {r['Name'][:110]}")
PY
done
