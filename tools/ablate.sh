#!/bin/bash
# Diagnostic: time the level-0 edge kernels (single-wavefront FC=2 kernel: SE3_NO_PAIR=1) with costs removed
# (results are wrong on purpose).  Mask bits: 1 GELU, 2 feature gather, 4 geometry gather, 8 stores.
set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
for v in ${@:-0 1 2 4 8 15}; do
  SE3_CXXFLAGS="-DSE3_ABLATE_MASK=$v" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "mask=$v: $(SE3_NO_PAIR=1 timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -A1 'level 0' | tail -1)"
done
