# A/B of the 3-byte row format of T / U (default; SE3_NO_T24=1 turns it off) on one box, alternating bench runs
set -u
mkdir -p gpurun_out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], s)'; }
for i in 1 2; do
  echo "[words]: $(SE3_NO_T24=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | line)"
  echo "[t24  ]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | line)"
done
