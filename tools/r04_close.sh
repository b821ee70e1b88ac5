#!/bin/bash
# round 4, closing evidence of the tree as it stands (one gpurun call): bench line + rocprofv3 kernel statistics + PMC passes
# + traffic file + errors (tools/round_profiles.sh), a second bench line once the traffic file of this tree is in place, every
# workload, per-level stage times, kernel statistics of the level-to-level convolutions, random-shape parity sweeps
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4close; mkdir -p $out
bash tools/round_profiles.sh r04f > $out/round_profiles.log 2>&1; echo "round_profiles rc=$?"
cp gpurun_out/prof_r04f/traffic.json profiles/r04_traffic.json 2>/dev/null   # (the copy on the box: bench.py reads it below)
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
bash tools/run_workloads.sh r4close_workloads > /dev/null 2>&1; cut -c1-300 gpurun_out/r4close_workloads/summary.txt
bash tools/levels_all.sh r4close_levels > /dev/null 2>&1
for w in headline dfaust_f2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/du_$w -o du -- python3 tools/profile_down_up.py --reps 10 --workload $w > $out/du_$w.log 2>&1
  cp $(find $out/du_$w -name "*kernel_stats.csv" | head -1) $out/down_up_${w}_kernel_stats.csv
done
timeout -k 10 500 python tools/fuzz_parity.py 48 20261006 > $out/fuzz.txt 2>&1; echo "fuzz rc=$? $(tail -1 $out/fuzz.txt)"
SE3_DX_PATH=1 timeout -k 10 500 python tools/fuzz_parity.py 48 20261007 > $out/fuzz_dx.txt 2>&1; echo "fuzz (edge-major dX wherever implemented) rc=$? $(tail -1 $out/fuzz_dx.txt)"
