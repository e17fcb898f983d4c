#!/bin/bash
# Development aid: build libvors_hip_e<TAG>.so with extra -D flags for BOTH LM objects (select with VORS_HIP_LIB=...).
# usage: tools/build_variant.sh TAG [-DFOO=1 ...]        (the other objects are taken from the last `make`)
set -e
TAG=$1; shift
CS=$(cd "$(dirname "$0")/../visual-odometry-rs_amd/csrc" && pwd)
cd $CS
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-c++20-extensions"
/opt/rocm/bin/hipcc $FLAGS -DVORS_FUSED=0 "$@" -c lm_kernels.hip -o /tmp/lme_$TAG.o &
/opt/rocm/bin/hipcc $FLAGS -DVORS_FUSED=1 "$@" -c lm_kernels.hip -o /tmp/lmf_$TAG.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill" | paste - - - | grep -E "error|split_eval_kernelILb0|lm_track_kernelILi256ELb0ELb1|lm_track_kernelILi128ELb0ELb0" | sed 's/remark: [^ ]*lm_kernels.hip:[0-9]*:[0-9]*: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[^ ]*lm_kernels.hip:[0-9]*:[0-9]*://g; s/Function Name: _ZN4vors//' | cut -c1-36,110-200
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC kernels.o /tmp/lme_$TAG.o /tmp/lmf_$TAG.o lm_reference.o dso_kernels.o capi.o multi.o -o ../vors_amd/libvors_hip_e$TAG.so -ldl -Wl,-rpath,/opt/rocm/lib
echo "built libvors_hip_e$TAG.so"
