#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_bounded_query.py tests/test_gpu_hierarchy.py tests/test_gpu_round3_boundary.py tests/test_gpu_concurrency.py -m gpu -q > $out/tests.log 2>&1
echo "tests rc=$?"; tail -6 $out/tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-fp32 --no-t16 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -c 400 $out/bench.err
python - <<PY
import json
r = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "e2e", r["end_to_end"]["ms_per_step"], "nbh", r["end_to_end"]["neighbourhood_ms"], "ov", r["end_to_end"]["overlapped"]["ms_per_step"])
for w in ("headline", "dfaust_f2"):
    for d in ("down", "up"):
        x = r["down_up"][w][d]; print(w, d, x["conv_only_ms"], x["with_neighbourhood_ms"], x["neighbourhood_and_transpose_ms"])
PY
SE3_BQ_TWO_PASS=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-fp32 --no-t16 > $out/bench2.json 2> $out/bench2.err
python - <<PY
import json
r = json.loads(open("$out/bench2.json").read().strip().splitlines()[-1])
print("TWO_PASS: e2e", r["end_to_end"]["ms_per_step"], "nbh", r["end_to_end"]["neighbourhood_ms"], "ov", r["end_to_end"]["overlapped"]["ms_per_step"])
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bq -- python3 tools/profile_bounded_bq.py 0 > $out/bq.log 2>&1
find $out -name "*kernel_stats*"
