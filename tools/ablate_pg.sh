# What bounds edge_param_grad_bf16_v2: diagnostic builds (wrong results), times from bench.py on one box.
# Bits of SE3_PG_ABLATE: 1 no GELU', 2 no feature gather, 4 no grad_T loads, 8 no d[A;beta] product.
set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
run() {
  SE3_CXXFLAGS="$1" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "[$1]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(s["edge_param_grad"])')"
}
for m in "$@"; do run "-DSE3_PG_ABLATE=$m"; done
