#!/bin/bash
# Development aid: build libvors_hip_d<TAG>.so with extra -D flags for dso_kernels.hip (select with VORS_HIP_LIB=...; tools/ab_dso.sh).
# usage: tools/build_dso_variant.sh TAG [-DFOO=1 ...]        (the other objects are taken from the last `make`)
set -e
TAG=$1; shift
CS=$(cd "$(dirname "$0")/../visual-odometry-rs_amd/csrc" && pwd)
cd $CS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -c dso_kernels.hip -o /tmp/dso_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC kernels.o lm_kernels.o lm_kernels_fused.o lm_reference.o /tmp/dso_$TAG.o capi.o multi.o -o ../vors_amd/libvors_hip_d$TAG.so -ldl -Wl,-rpath,/opt/rocm/lib
echo "built libvors_hip_d$TAG.so"
