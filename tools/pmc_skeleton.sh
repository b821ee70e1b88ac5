#!/bin/bash
# PMC counters of the edge kernel skeleton (SE3_ABLATE_MASK=15) -- diagnostic
set -u
export SE3_LIB_SUFFIX=_ab  # variant builds go to lib/libse3conv_hip_ab.so (se3conv3d_amd/build.py): the shipped library is never overwritten
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SE3_CXXFLAGS="-DSE3_ABLATE_MASK=${1:-15}" python -m se3conv3d_amd.build --force > /dev/null 2>&1
out=gpurun_out/pmc_skel
mkdir -p $out
export SE3_NO_PAIR=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out -o sq1 -- python3 tools/profile_layer.py > $out/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_SCA --output-format csv -d $out -o sq2 -- python3 tools/profile_layer.py > $out/sq2.log 2>&1
