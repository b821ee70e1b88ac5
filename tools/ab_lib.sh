#!/bin/bash
# A/B of the shipped library against variant builds: tools/ab_lib.sh <outdir> <suffix> [<suffix> ...]  ("-" = shipped)
set -u
out=gpurun_out/$1; shift
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], {k:s[k] for k in s if k.startswith("edge")})'; }
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e="SE3_AB=0"; else e="SE3_LIB_SUFFIX=$v"; fi
  echo "[$v]: $(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 30 2>&1 | line)"
done
done | tee $out/ab.log
