#!/bin/bash
# Evidence of a round in two gpurun calls (usage: tools/round_close.sh <tag> 1|2):
#   1: tools/round_profiles.sh (bench line, rocprofv3 kernel statistics of bench.py and of the level-0 layer, 4 --pmc passes,
#      traffic file with the hash of the kernel sources, errors against the fp64 oracle), every workload, every level
#   2: (after copying gpurun_out/prof_<tag>/traffic.json to profiles/<tag>_traffic.json) bench.py with that file in place
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${2:-1}" = "1" ]; then
  bash tools/round_profiles.sh $tag
  bash tools/run_workloads.sh workloads_$tag
  bash tools/levels_all.sh levels_$tag > /dev/null
  tail -40 gpurun_out/levels_$tag/levels.txt
else
  mkdir -p gpurun_out/prof_$tag
  timeout -k 10 500 python bench.py > gpurun_out/prof_$tag/bench_final.json 2> gpurun_out/prof_$tag/bench_final.err
  echo "bench rc=$?"; tail -c 400 gpurun_out/prof_$tag/bench_final.err
  python -c "
import json
r = json.loads(open('gpurun_out/prof_$tag/bench_final.json').read().strip().splitlines()[-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'layer', r['single_layer']['ms_per_step'], 'traffic', r['roofline']['traffic'], r['roofline']['traffic_source'][:60])
print('roofline', {k: r['roofline'][k] for k in ('kernel','achieved','frac','avg_launch_ms')})
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
"
fi
