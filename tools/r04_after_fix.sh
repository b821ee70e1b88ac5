#!/bin/bash
# round 4: after the memset-free transposition -- the GPU suite, the one-rank RCCL rehearsal of the whole bench line (with the
# level-to-level legs captured), a plain bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4x; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/tests.log 2>&1; echo "tests rc=$? $(tail -1 $out/tests.log)"
SE3_BENCH_VERBOSE=1 SE3_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 timeout -k 10 500 python bench.py --no-cpu-baseline > $out/bench_dist.json 2> $out/bench_dist.err; echo "bench next to a communicator rc=$? $(grep -c 'Memory access fault' $out/bench_dist.err)"
timeout -k 10 500 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench_dist", "bench"):
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
