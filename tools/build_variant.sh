#!/bin/bash
# Development aid: build libvors_hip_e<TAG>.so with extra -D flags for the FUSED object (select with VORS_HIP_LIB=...).
# usage: tools/build_variant.sh TAG [-DFOO=1 ...]
set -e
TAG=$1; shift
CS=/root/repo/visual-odometry-rs_amd/csrc
cd $CS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DVORS_FUSED=1 "$@" -c lm_kernels.hip -o /tmp/lmf_$TAG.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill" | paste - - - | grep -E "error|split_eval_kernelILb0|lm_track_kernelILi256ELb0ELb1" | sed 's/remark: [^ ]*lm_kernels.hip:[0-9]*:[0-9]*: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/[^ ]*lm_kernels.hip:[0-9]*:[0-9]*://g; s/Function Name: _ZN4vors//' | cut -c1-36,110-200
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC kernels.o lm_kernels.o /tmp/lmf_$TAG.o dso_kernels.o capi.o multi.o -o ../vors_amd/libvors_hip_e$TAG.so -ldl -Wl,-rpath,/opt/rocm/lib
