#!/bin/bash
# Measurement build with the per-step ablation switches of igemm_v4_kernel (SPX_V4_DBG, see igemm.hip):
# lib/libspconv_amd_abl.so, selected with SPX_LIB.  Results of ablated runs are wrong by design.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT/abl
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -DSPX_ABLATE"
$HIPCC $FLAGS -c igemm.hip -o $OUT/abl/igemm.o
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libspconv_amd_abl.so $OUT/rulebook.o $OUT/abl/igemm.o $OUT/pool.o $OUT/tileplan.o $OUT/common.o
echo built $OUT/libspconv_amd_abl.so
