#!/bin/bash
# round 4, first GPU pass: full GPU suite, the bench line (with the down_up leg), kernel statistics of the level-to-level convs
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4a
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
echo "tests rc=$?" | tee $out/tests.rc
tail -5 $out/tests.log
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err && tail -c 1500 $out/bench.json
for w in headline dfaust_f2; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o down_up_$w -- python3 tools/profile_down_up.py --workload $w --reps 10 > $out/down_up_$w.log 2>&1
done
find $out -name "*kernel_stats*"
