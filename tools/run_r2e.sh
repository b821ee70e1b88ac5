#!/bin/bash
set -u
out=gpurun_out/${1:-r2e}
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_glue.py -q > $out/tests_glue.log 2>&1; echo "glue rc=$?"; tail -3 $out/tests_glue.log
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_backward.py -x -q > $out/tests_full.log 2>&1; echo "fullsize rc=$?"; tail -5 $out/tests_full.log
timeout -k 10 300 python tools/time_block.py > $out/block.log 2>&1; grep "^level" $out/block.log
