#!/bin/bash
# round 5, second GPU pass: the new test files, the sliced-schedule variant in the variants test, the slicing A/B on the
# scene workload the verdict names, and a first full bench line of the round.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_second
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_network_replay.py tests/test_gpu_graph_nodes.py -m gpu -q -s -p no:cacheprovider > $out/new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 $out/new_tests.log
SLICE="golden or random_shapes or headline_subset or features_only or empty_rows"
SE3_SLICE_MB=1 SE3_SLICE_STREAMS=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SLICE" -p no:cacheprovider > $out/parity_sliced.log 2>&1
echo "parity sliced rc=$?"; tail -3 $out/parity_sliced.log
timeout -k 10 900 python -m pytest tests/test_gpu_down_up.py tests/test_gpu_transpose.py tests/test_gpu_concurrency.py -m gpu -q -p no:cacheprovider > $out/more_tests.log 2>&1
echo "more tests rc=$?"; tail -4 $out/more_tests.log
bash tools/ab.sh r05_second --reps 2 --bench-args "--no-cpu-baseline --no-extra --steps 30 --workload scannet150k_f1" - \
  env:SE3_SLICE_MB=96 env:SE3_SLICE_MB=96,SE3_SLICE_STREAMS=2 env:SE3_SLICE_MB=128,SE3_SLICE_STREAMS=2 env:SE3_SLICE_MB=64,SE3_SLICE_STREAMS=2 env:SE3_SLICE_MB=32
