set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
run() {
  SE3_CXXFLAGS="$1" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "[$1]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(j["single_layer"]["ms_per_step"], j["roofline"]["stages_ms"]["gemm_gradT"])')"
}
run "-DSE3_STRIP_ROT=17"
run "-DSE3_STRIP_ROT=17 -DSE3_STRIP_NOSTORE"
run "-DSE3_STRIP_ROT=1"
run "-DSE3_STRIP_ROT=5"
run "-DSE3_STRIP_ROT=0"
