#!/bin/bash
# bench.py on every workload of se3conv3d_amd/workloads.py (1 GPU): one summary line each.
set -u
out=gpurun_out/${1:-workloads}
mkdir -p $out
for w in headline scannet150k_f1 dfaust_f2 dfaust_f4; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > $out/$w.json 2> $out/$w.err
  echo "[$w] rc=$? $(python -c "
import json,sys
try:
    d=json.loads([l for l in open('$out/$w.json') if l.startswith('{')][-1])
    c=d['config']; r=d.get('roofline') or {}
    print('value',d['value'],'ms',d['ms_per_step'],'levels',c['level_points'],'edges',c['level_edges'],'layer_ms',d['single_layer']['ms_per_step'],'layer_frac',d['layer_frac'],'stack_frac',d['stack_frac'],'dominant',r.get('kernel'),r.get('frac'),'fp32',d.get('fp32_mode',{}).get('ms_per_step'),'e2e',d['end_to_end']['ms_per_step'])
except Exception as e:
    print('FAILED', e); print(open('$out/$w.err').read()[-1500:])
")"
done | tee $out/summary.txt
