# What bounds the wave-pair edge kernel: diagnostic builds with ingredients removed (results are wrong on purpose),
# edge_t_fwd / edge_t_transposed times from bench.py on one box.  Bits of SE3_PAIR_ABLATE: 1 no GELU, 2 no feature
# gather, 4 no T stores, 8 no hi/lo split of phi, 16 neighbour ids computed instead of loaded, 32 row extents computed
# instead of loaded (31 edges per point).
set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
run() {
  SE3_CXXFLAGS="$1" python -m se3conv3d_amd.build --force > /dev/null 2>&1
  echo "[$1]: $(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | python -c 'import sys,json; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=j["roofline"]["stages_ms"]; print({k:s[k] for k in s if k.startswith("edge_t")})')"
}
for m in "$@"; do run "-DSE3_PAIR_ABLATE=$m"; done
