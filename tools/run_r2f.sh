#!/bin/bash
set -u
out=gpurun_out/${1:-r2f}
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_bounded_query.py tests/test_gpu_negative_zero.py tests/test_gpu_concurrency.py -q > $out/tests_a.log 2>&1; echo "bounded/negzero/conc rc=$?"; tail -3 $out/tests_a.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "other_basis or golden or random_shapes or empty" > $out/tests_b.log 2>&1; echo "parity slice rc=$?"; tail -3 $out/tests_b.log
for v in "SE3CONV_FUSED=0" "SE3CONV_FUSED_ROWS=4096" "SE3CONV_FUSED_ROWS=40000"; do
  echo "== $v"; env $v timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -v amdgpu.ids
done | tee $out/levels.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1]); print({k:d[k] for k in ('value','ms_per_step','layer_frac','stack_frac')}, d['single_layer']['ms_per_step'], d['eager'], d['end_to_end'], d.get('fp32_mode'))"
