#!/bin/bash
# round 5, third GPU pass: fused-tile contraction probe, the edge kernel at capped occupancy, graph-node and concurrency tests
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_third
mkdir -p $out
timeout -k 10 120 tools/probes/fused_tile > $out/fused_tile.txt 2>&1
echo "fused_tile rc=$?"; cat $out/fused_tile.txt
timeout -k 10 600 python -m pytest tests/test_gpu_graph_nodes.py tests/test_gpu_concurrency.py -m gpu -q -s -p no:cacheprovider > $out/tests.log 2>&1
echo "tests rc=$?"; grep -a "memset nodes" $out/tests.log; tail -5 $out/tests.log
bash tools/ab.sh r05_third --reps 2 - env:SE3_PAIR_OCC=3 env:SE3_PAIR_OCC=2 env:SE3_PAIR_OCC=1
