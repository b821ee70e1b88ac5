#!/bin/bash
# targeted GPU pass: $1 = output tag, $2 = pytest -k expression over the parity files ("" = skip), $3 = "bench" to run bench.py
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$1
mkdir -p $out
if [ -n "${2:-}" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_backward.py tests/test_gpu_down_up.py -m gpu -q -k "$2" > $out/tests.log 2>&1
  echo "tests rc=$?" | tee $out/tests.rc
  tail -4 $out/tests.log
fi
if [ "${3:-}" = "bench" ]; then
  timeout -k 10 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
  echo "bench rc=$?"
  tail -c 600 $out/bench.err
  python - <<PY
import json
try:
    r = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "layer", r["single_layer"]["ms_per_step"])
    print("stages", r["roofline"]["stages_ms"])
    print("t16", json.dumps(r.get("t16_mode"))[:1500])
    print("e2e", r["end_to_end"]["ms_per_step"], "fp32", r.get("fp32_mode", {}).get("ms_per_step"))
except Exception as exc:
    print("no bench line:", exc)
PY
fi
