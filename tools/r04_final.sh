#!/bin/bash
# round 4, closing measurements of the final tree: bench line (traffic file in place), every workload in the default and the
# T16 mode, per-level stage times, random-shape parity sweep over the three arithmetic modes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4u
mkdir -p $out
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
bash tools/run_workloads.sh r4u_workloads > /dev/null 2>&1; cat gpurun_out/r4u_workloads/summary.txt | cut -c1-400
for w in headline scannet150k_f1 dfaust_f2 dfaust_f4; do
  timeout -k 10 300 python bench.py --workload $w --precision bf16x3_t16 --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $out/t16_$w.json 2> $out/t16_$w.err
  echo "[t16 $w] $(python -c "
import json
d=json.loads([l for l in open('$out/t16_$w.json') if l.startswith('{')][-1]); print('value',d['value'],'ms',d['ms_per_step'],'layer_ms',d['single_layer']['ms_per_step'])")"
done | tee $out/t16_workloads.txt
bash tools/levels_all.sh r4u_levels > /dev/null 2>&1; cp gpurun_out/r4u_levels/levels.txt $out/levels.txt
timeout -k 10 600 python tools/fuzz_parity.py 72 20261005 > $out/fuzz.txt 2>&1; tail -3 $out/fuzz.txt
