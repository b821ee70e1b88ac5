#!/bin/bash
# A/B of the 19 KB ("lean") wave-pair parameter-gradient kernel against the 26 KB form (variant library _nolean)
set -u
out=gpurun_out/${1:-ab_pglean}
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], {k:s[k] for k in s if k.startswith("edge")})'; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_backward.py -x -q -k "not beyond" > $out/tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/tests.log
for v in "SE3_PG_PAIR_WGS=8" "SE3_LIB_SUFFIX=_nolean" "SE3_PG_PAIR_WGS=7" "SE3_PG_PAIR_WGS=6" "SE3_PG_PAIR_WGS=8" "SE3_LIB_SUFFIX=_nolean"; do
  echo "[$v]: $(env $v timeout -k 10 200 python bench.py --no-cpu-baseline --no-fp32 --steps 20 2>&1 | line)"
done | tee $out/ab.log
