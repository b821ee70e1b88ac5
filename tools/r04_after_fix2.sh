#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4x2; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_down_up.py tests/test_gpu_bounded_query.py tests/test_gpu_parity.py -m gpu -q -x > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"; grep -m3 "Error\|assert " $out/tests.log
bash tools/r04_fault_bisect.sh tr full
timeout -k 10 500 python bench.py --no-cpu-baseline --no-fp32 --no-t16 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench",):
    try:
        r = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", r["value"], "ms", r["ms_per_step"], "layer", r["single_layer"]["ms_per_step"], "e2e", r["end_to_end"]["ms_per_step"], r["end_to_end"]["overlapped"]["ms_per_step"])
        for w, leg in r["down_up"].items():
            if isinstance(leg, dict):
                for n, v in leg.items():
                    print("  ", w, n, v["launch"], v["conv_only_ms"], v["with_neighbourhood_ms"], v["neighbourhood_and_transpose_ms"])
    except Exception as exc:
        print(f, "no line:", exc)
PY
