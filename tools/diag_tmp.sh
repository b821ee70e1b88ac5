line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], s)'; }
for fl in "-DSE3_NT_STORES=0" "-DSE3_NT_STORES=1" "-DSE3_NT_STORES=0" "-DSE3_NT_STORES=1"; do
  SE3_CXXFLAGS="$fl" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "[$fl]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>&1 | line)"
done
python -m se3conv3d_amd.build --force > /dev/null 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
