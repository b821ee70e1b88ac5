#!/bin/bash
set -u
out=gpurun_out/${1:-ab_stream3}
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], {k:s[k] for k in s if k.startswith("edge")})'; }
timeout -k 10 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_fullsize_backward.py -x -q > $out/tests.log 2>&1
echo "tests rc=$?"; tail -5 $out/tests.log
SE3_LIB_SUFFIX=_w5 SE3_PAIR_STREAM=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or random_shapes or headline_subset" > $out/tests_w5.log 2>&1
echo "w5 parity rc=$?"; tail -2 $out/tests_w5.log
for v in "SE3_PAIR_STREAM=0" "SE3_PAIR_STREAM=1" "SE3_LIB_SUFFIX=_w5 SE3_PAIR_STREAM=1" "SE3_LIB_SUFFIX=_w5 SE3_PAIR_STREAM=1 SE3_PAIR_STREAM_WGS=8" "SE3_LIB_SUFFIX=_w5 SE3_PAIR_STREAM=1 SE3_PAIR_STREAM_WGS=12"; do
  echo "[$v]: $(env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 20 2>&1 | line)"
done | tee $out/ab.log
timeout -k 10 300 python tools/time_block.py > $out/block.log 2>&1; grep "^level" $out/block.log
