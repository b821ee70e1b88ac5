#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_dbg
mkdir -p $out
export SE3_BENCH_VERBOSE=1
timeout -k 10 400 python -X faulthandler bench.py --no-cpu-baseline > $out/bench_plain.out 2> $out/bench_plain.err
echo "plain rc=$?"
tail -c 600 $out/bench_plain.err
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r05_dbg/bench_plain.out").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "layer", r["single_layer"]["ms_per_step"])
    print("levels_concurrent", r.get("levels_concurrent"))
    print("fwd", r.get("forward_only"), "e2e", r["end_to_end"]["ms_per_step"], r["end_to_end"]["overlapped"]["ms_per_step"])
    print("down_up", {k: (v["conv_only_ms"], v["with_neighbourhood_ms"]) for k, v in r["down_up"]["headline"].items()})
    print("fp32", r.get("fp32_mode", {}).get("ms_per_step"), "t16", r.get("t16_mode", {}).get("ms_per_step"))
except Exception as e:
    print("no line", e)
PY
bash tools/gpu_suite.sh r05_suite_b
