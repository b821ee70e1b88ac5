#!/bin/bash
# round 5, first GPU pass of the row-sliced schedules: parity with slicing on (small slices so that the golden / random shapes
# are sliced too, full-size backward at 96 MB), then the A/B of slice sizes x streams against the whole-tensor schedule.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_slice_first
mkdir -p $out
SLICE="golden or random_shapes or headline_subset or features_only or empty_rows"
for st in 1 2; do
  SE3_SLICE_MB=1 SE3_SLICE_STREAMS=$st timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SLICE" -p no:cacheprovider > $out/parity_s$st.log 2>&1
  echo "parity slice streams=$st rc=$?"; tail -3 $out/parity_s$st.log
done
SE3_SLICE_MB=96 SE3_SLICE_STREAMS=2 timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_backward.py -m gpu -x -q -p no:cacheprovider > $out/fullsize.log 2>&1
echo "fullsize rc=$?"; tail -3 $out/fullsize.log
bash tools/ab.sh r05_slice_first --reps 2 - \
  env:SE3_SLICE_MB=96 env:SE3_SLICE_MB=96,SE3_SLICE_STREAMS=2 \
  env:SE3_SLICE_MB=64 env:SE3_SLICE_MB=64,SE3_SLICE_STREAMS=2 \
  env:SE3_SLICE_MB=128 env:SE3_SLICE_MB=128,SE3_SLICE_STREAMS=2 \
  env:SE3_SLICE_MB=32,SE3_SLICE_STREAMS=2 env:SE3_SLICE_MB=200,SE3_SLICE_STREAMS=2
