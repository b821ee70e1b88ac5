#!/bin/bash
# the whole -m gpu suite as the driver runs it, with per-test durations (what tests/ costs on the GPU box)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-gpu_suite}
mkdir -p $out
start=$(date +%s)
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=60 -p no:cacheprovider > $out/suite.log 2>&1
echo "suite rc=$? wall=$(( $(date +%s) - start )) s"
grep -a -E "passed|failed|error" $out/suite.log | tail -3
grep -a -A62 "slowest 60 durations" $out/suite.log | head -70
