# Codegen variants of the edge kernels (run on the GPU box): SLP vectorizer (packed fp32 VALU) on/off, packed GELU.
set -u
run() {
  SE3_CXXFLAGS="$1" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "[$1]: $(timeout -k 10 200 python tools/profile_levels.py 2>&1 | grep -A1 'level 0' | tail -1)"
}
run "-fno-slp-vectorize"
run ""
run "-DSE3_GELU_PK=1 -DSE3_PAIR_WAVES=3"
python -m se3conv3d_amd.build --force > /dev/null 2>&1
