#!/bin/bash
# full GPU suite (no -x: every failure of the pass is wanted), log under gpurun_out/$1
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q ${2:-} > $out/tests.log 2>&1
echo "tests rc=$?" | tee $out/tests.rc
tail -15 $out/tests.log
