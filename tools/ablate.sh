#!/bin/bash
# Diagnostic: time the level-0 edge kernels with one cost removed at a time (results are wrong on purpose).
# Run on the GPU box: rebuilds the library per variant into the box's scratch copy.
set -u
for v in 0 1 2 3 4; do
  SE3_CXXFLAGS="-DSE3_ABLATE=$v" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "ablate=$v: $(timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -A1 'level 0' | tail -1)"
done
python -m se3conv3d_amd.build --force > /dev/null 2>&1
