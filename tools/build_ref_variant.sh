#!/bin/bash
# Development aid: build libvors_hip_r<TAG>.so with extra -D flags for lm_reference.hip (select with VORS_HIP_LIB=...).
# usage: tools/build_ref_variant.sh TAG [-DVORS_REFW_TIMING ...]        (the other objects are taken from the last `make`)
set -e
TAG=$1; shift
CS=$(cd "$(dirname "$0")/../visual-odometry-rs_amd/csrc" && pwd)
cd $CS
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c lm_reference.hip -o /tmp/lmr_$TAG.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|LDS Size" | sed 's/.*remark: [^ ]* //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - | grep -E "error|lm_ref_track" | sed 's/Function Name: _ZN4vors//' | cut -c1-40,120-260
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC kernels.o lm_kernels.o lm_kernels_fused.o /tmp/lmr_$TAG.o dso_kernels.o capi.o multi.o -o ../vors_amd/libvors_hip_r$TAG.so -ldl -Wl,-rpath,/opt/rocm/lib
echo "built libvors_hip_r$TAG.so"
