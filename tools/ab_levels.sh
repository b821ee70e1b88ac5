#!/bin/bash
# graph-replay time per level of the bench stack under environment switches: tools/ab_levels.sh <outdir> "<VAR=val>" ... ("-" = defaults)
set -u
out=gpurun_out/$1; shift
mkdir -p $out
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "[$v]: $(env $e timeout -k 10 200 python tools/profile_levels.py 2>/dev/null | grep 'graph replay' | sed 's/.*n *\([0-9]*\) rows.*graph replay \([0-9.]*\) ms.*/\1:\2/' | tr '\n' ' ')"
done
done | tee $out/ab.log
