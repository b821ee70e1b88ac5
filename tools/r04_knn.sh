#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4knn; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "knn or frames or pca or hierarchy or boundary" > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"; grep -m5 "Error\|FAILED" $out/tests.log | cut -c1-300
timeout -k 10 300 python tools/time_frames.py 2>&1 | grep -v amdgpu.ids | cut -c1-260
