#!/bin/bash
# One A/B harness for every switch of the library (run on the GPU box through gpurun), replacing the ab_*.sh /
# ablate*.sh / variants_*.sh family:
#
#   tools/ab.sh <outdir> [--reps N] [--levels] [--bench-args "<args>"] <variant> [<variant> ...]
#
# A variant is "-" (the shipped library, default environment), "lib:<suffix>" (a variant build
# lib/libse3conv_hip<suffix>.so made beforehand with SE3_LIB_SUFFIX=<suffix> SE3_CXXFLAGS="-D..." python -m
# se3conv3d_amd.build -- cross-compiled in the build container, it travels with the snapshot), "env:VAR=val,VAR2=val"
# (environment switches), or "lib:<suffix>+env:VAR=val".  Variants alternate inside every repetition, so box-to-box
# and drift effects cancel.  One line per run: ms per step, single-layer ms, every stage time of the full-resolution
# layer; with --levels the graph-replay time of every level of the stack instead (tools/profile_levels.py).
set -u
out=gpurun_out/$1; shift
reps=3; levels=0; bargs="--no-cpu-baseline --no-extra --steps 30"
while [ $# -gt 0 ]; do
  case "$1" in
    --reps) reps=$2; shift 2;;
    --levels) levels=1; shift;;
    --bench-args) bargs="$2"; shift 2;;
    *) break;;
  esac
done
mkdir -p $out
line() { python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(j["ms_per_step"], j["single_layer"]["ms_per_step"], json.dumps(j["roofline"]["stages_ms"]))'; }
for rep in $(seq 1 $reps); do
for v in "$@"; do
  e=""
  for part in ${v//+/ }; do
    case "$part" in
      -) ;;
      lib:*) e="$e SE3_LIB_SUFFIX=${part#lib:}";;
      env:*) e="$e ${part#env:}"; e="${e//,/ }";;
      *) echo "bad variant $part"; exit 2;;
    esac
  done
  if [ $levels = 1 ]; then
    echo "[$v]: $(env $e timeout -k 10 200 python tools/profile_levels.py 2>/dev/null | grep 'graph replay' | sed 's/.*n *\([0-9]*\) rows.*graph replay \([0-9.]*\) ms.*/\1:\2/' | tr '\n' ' ')"
  else
    echo "[$v]: $(env $e timeout -k 10 200 python bench.py $bargs 2>&1 | line)"
  fi
done
done | tee $out/ab.log
