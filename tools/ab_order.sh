#!/bin/bash
# row order of the input points: as drawn (the headline) against Z-order sorted per scene
set -u
out=gpurun_out/${1:-ab_order}
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], j.get("end_to_end",{}).get("ms_per_step"), s)'; }
for v in "random" "morton" "random" "morton"; do
  echo "[$v]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 20 --point-order $v 2>&1 | line)"
done | tee $out/ab.log
for w in scannet150k_f1 dfaust_f2; do
  echo "[$w morton]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 20 --workload $w --point-order morton 2>&1 | line)"
done | tee -a $out/ab.log
