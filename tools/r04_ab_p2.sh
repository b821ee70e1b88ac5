#!/bin/bash
# round 4: (1) the transposition without rocPRIM's one-sweep sort: parity tests that use it, the fault bisect next to a live
# communicator; (2) two-tile wave-pair kernel at 2 instead of 3 wavefronts per SIMD (no spills) on the wide levels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4w
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_down_up.py tests/test_gpu_bounded_query.py tests/test_gpu_hierarchy.py -m gpu -q -x > gpurun_out/r4w/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r4w/tests.log)"
bash tools/r04_fault_bisect.sh tr full
bash tools/ab.sh r4w_p2 --reps 2 --bench-args "--workload dfaust_f2 --no-cpu-baseline --no-extra --steps 30" - lib:_p2w2
for v in "" _p2w2; do echo "shapes [$v]"; SE3_LIB_SUFFIX=$v timeout -k 10 200 python tools/bench_shapes.py 2>&1 | grep -i "wide\|headline"; done
