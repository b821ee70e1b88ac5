#!/bin/bash
# round 4 A/B pass 1: the edge kernels' VMEM diet (gather shape upper bound, dynamic item claiming) and the T16 mode's parts
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab.sh r4g_edge --reps 3 - lib:_g128 env:SE3_PAIR_PERSIST=2048 env:SE3_PAIR_PERSIST=2048,SE3_PAIR_DYNAMIC=1 env:SE3_PAIR_PERSIST=4096,SE3_PAIR_DYNAMIC=1
bash tools/ab.sh r4g_t16 --reps 2 --bench-args "--no-cpu-baseline --no-extra --steps 30 --precision bf16x3_t16" - env:SE3_T16_GT=1 lib:_t16a1 lib:_t16a2 lib:_t16a4 lib:_t16a7
bash tools/ab.sh r4g_base --reps 2 -
