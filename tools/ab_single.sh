#!/bin/bash
set -u
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], {k:s[k] for k in s if k.startswith("edge")})'; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "scannet or random_shapes or golden" 2>&1 | tail -2
for v in "SE3_LIB_SUFFIX=" "SE3_LIB_SUFFIX=_s3" "SE3_LIB_SUFFIX=" "SE3_LIB_SUFFIX=_s3"; do
  echo "[$v]: $(env $v timeout -k 10 200 python bench.py --workload scannet150k_f1 --no-cpu-baseline --no-fp32 --steps 20 2>&1 | line)"
done
