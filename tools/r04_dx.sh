#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4dx; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_transpose.py tests/test_gpu_variants.py tests/test_gpu_down_up.py tests/test_gpu_bounded_query.py tests/test_gpu_parity.py -m gpu -q -x -k "${1:-DX_PATH or down or up or two_clouds or transpose or csr}" > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"; grep -m5 "Error\|assert \|rel err\|FAILED" $out/tests.log | cut -c1-300
timeout -k 10 500 python bench.py --no-cpu-baseline --no-fp32 --no-t16 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    r = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("value", r["value"], "ms", r["ms_per_step"], "layer", r["single_layer"]["ms_per_step"], "e2e", r["end_to_end"]["ms_per_step"], r["end_to_end"]["overlapped"]["ms_per_step"])
    for w, leg in r["down_up"].items():
        if isinstance(leg, dict):
            for n, v in leg.items():
                print("  ", w, n, v["launch"], v["conv_only_ms"], v["with_neighbourhood_ms"], v["neighbourhood_and_transpose_ms"], v["stages_ms"])
except Exception as exc:
    print("no line:", exc); print(open("$out/bench.err").read()[-1500:])
PY
