# usage: tools/ab_env.sh "VAR=value" [runs]: alternating bench runs with / without an environment setting, one box
set -u
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], s)'; }
for i in $(seq 1 ${2:-2}); do
  echo "[default ]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>&1 | line)"
  echo "[$1]: $(env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>&1 | line)"
done
