#!/bin/bash
# round 4 (second session): one summary line per workload and the per-level stage times of the tree as it stands
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/run_workloads.sh r4v_workloads > /dev/null 2>&1; cut -c1-420 gpurun_out/r4v_workloads/summary.txt
bash tools/levels_all.sh r4v_levels > /dev/null 2>&1; tail -40 gpurun_out/r4v_levels/levels.txt
