#!/bin/bash
# per-level stage times of every workload (SE3_OVERLAP_ROWS=0: branches back to back, so stage events are clean)
set -u
out=gpurun_out/${1:-levels}
mkdir -p $out
for w in headline dfaust_f2 dfaust_f4 scannet150k_f1; do
  echo "== $w" | tee -a $out/levels.txt
  SE3_OVERLAP_ROWS=0 timeout -k 10 200 python tools/profile_levels.py $w >> $out/levels.txt 2>> $out/levels.err || echo "rc=$?" >> $out/levels.txt
done
cat $out/levels.txt
