#!/bin/bash
# Development aid: build libvors_hip_k<TAG>.so with extra -D flags for kernels.hip (select with VORS_HIP_LIB=...).
# usage: tools/build_kf_variant.sh TAG [-DVORS_KF_BATCH=1 ...]        (the other objects are taken from the last `make`)
set -e
TAG=$1; shift
CS=$(cd "$(dirname "$0")/../visual-odometry-rs_amd/csrc" && pwd)
cd $CS
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c kernels.hip -o /tmp/k_$TAG.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - | grep -E "error|keyframe_sparse" | sed 's/Function Name: _ZN4vors//' | cut -c1-40,100-300
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/k_$TAG.o lm_kernels.o lm_kernels_fused.o lm_reference.o dso_kernels.o capi.o multi.o -o ../vors_amd/libvors_hip_k$TAG.so -ldl -Wl,-rpath,/opt/rocm/lib
echo "built libvors_hip_k$TAG.so"
