#!/bin/bash
# Measurement builds of igemm.hip with ONE part of a gather-GEMM step compiled out (SPX_ABL in igemm.hip:
# 1 = weights staged once, 2 = also no per-step barrier, 3 = no MFMAs, 4 = no gathered-row loads, 5 = no
# pair-word loads; 0 = nothing removed): lib/libspconv_amd_abl<N>.so, selected with SPX_LIB.  Results of
# ablated runs are wrong by design; tools/dense_probe.py times them.  (f16 kernels only: the other operand types are
# linked from the product build, csrc/build.sh first.)
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT/abl
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16"
for v in ${@:-0 1 2 3 4 5}; do
  ( $HIPCC $FLAGS -DSPX_ABLATE=$v -c igemm.hip -o $OUT/abl/igemm$v.o &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libspconv_amd_abl$v.so $OUT/rulebook.o $OUT/abl/igemm$v.o $OUT/igemm_bf16.o $OUT/igemm_f32.o $OUT/igemm_i8.o $OUT/igemm_gen1.o $OUT/igemm_ws.o $OUT/igemm_bwdn.o $OUT/pool.o $OUT/rowsort.o $OUT/norm.o $OUT/common.o &&
    echo built $OUT/libspconv_amd_abl$v.so ) &
done
wait
