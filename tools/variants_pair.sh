# A/B of edge-kernel build variants on one box (run through gpurun): two bench runs per variant.
set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
run() {
  SE3_CXXFLAGS="$1" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  for i in 1 2; do
  echo "[$1]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print(j["ms_per_step"], j["single_layer"]["ms_per_step"], {k:s[k] for k in s if k.startswith("edge")})')"
  done
}
for v in "$@"; do run "$v"; done
