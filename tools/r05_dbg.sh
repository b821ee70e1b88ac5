#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab.sh r05_kg_levels --reps 3 --levels - env:SE3_NN_KG=1
for w in headline dfaust_f2 dfaust_f4 scannet150k_f1; do
  echo "== $w"
  bash tools/ab.sh r05_kg_$w --reps 3 --bench-args "--no-cpu-baseline --no-extra --steps 30 --workload $w" - env:SE3_NN_KG=1 | cut -c1-50
done
timeout -k 10 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fullsize_backward.py -m gpu -q -x -p no:cacheprovider --durations=8 2>&1 | tail -14
