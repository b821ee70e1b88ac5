#!/bin/bash
# rocprofv3 counter passes for tools/profile_layer.py (run on the GPU box through gpurun).
# usage: tools/pmc_passes.sh <tag> [extra args for profile_layer.py]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -o $name -- python3 tools/profile_layer.py "${EXTRA[@]}" > $out/$name.log 2>&1 || echo "pass $name failed"
}
EXTRA=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls $out
