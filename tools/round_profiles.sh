#!/bin/bash
# Evidence run for profiles/ (on the GPU box through gpurun): bench line, kernel-trace stats of the same program,
# PMC passes of one full-resolution layer.   usage: tools/round_profiles.sh <tag>
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_$tag
timeout -k 10 400 python bench.py > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --no-graph > gpurun_out/prof_$tag/stats.log 2>&1
# the same statistics for the full-resolution layer alone: per-kernel averages there are level-0 launch times
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o layer -- python3 tools/profile_layer.py --reps 10 > gpurun_out/prof_$tag/layer.log 2>&1
bash tools/pmc_passes.sh $tag > gpurun_out/prof_$tag/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/prof_$tag/pmc_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$tag gpurun_out/prof_$tag/traffic.json > /dev/null 2>&1
python tools/report_errors.py > gpurun_out/prof_$tag/errors.txt 2>&1
tail -1 gpurun_out/prof_$tag/bench.json
find gpurun_out/prof_$tag -name "*kernel_stats*" | head
