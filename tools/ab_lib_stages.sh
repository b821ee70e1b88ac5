#!/bin/bash
# as tools/ab_lib.sh, printing every stage time:  tools/ab_lib_stages.sh <outdir> <suffix> ...   ("-" = shipped)
set -u
out=gpurun_out/$1; shift
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(j["ms_per_step"], j["single_layer"]["ms_per_step"], j["roofline"]["stages_ms"])'; }
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e="SE3_AB=0"; else e="SE3_LIB_SUFFIX=$v"; fi
  echo "[$v]: $(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>&1 | line)"
done
done | tee $out/ab.log
