#!/bin/bash
set -u
out=gpurun_out/r2g
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests_all.log 2>&1; echo "all gpu tests rc=$?"; tail -4 $out/tests_all.log
bash tools/round_profiles.sh r02a > $out/profiles.log 2>&1; echo "profiles rc=$?"; tail -3 $out/profiles.log
