#!/bin/bash
# generic A/B of environment switches: tools/ab_env.sh <outdir> "<VAR=val ...>" "<VAR=val ...>" ...   ("-" = defaults)
set -u
out=gpurun_out/$1; shift
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(j["ms_per_step"], j["single_layer"]["ms_per_step"], j.get("eager",{}).get("ms_per_step"), j.get("end_to_end",{}).get("ms_per_step"))'; }
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "[$v]: $(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>&1 | line)"
done
done | tee $out/ab.log
