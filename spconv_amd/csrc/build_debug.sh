#!/bin/bash
# Debug build with the in-kernel timeline stamps (tools/timeline.py): lib/libspconv_amd_dbg.so
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT/dbg
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -DSPX_TIMELINE"
$HIPCC $FLAGS -c rulebook.hip -o $OUT/dbg/rulebook.o &
$HIPCC $FLAGS -c igemm.hip -o $OUT/dbg/igemm.o &
$HIPCC $FLAGS -c pool.hip -o $OUT/dbg/pool.o &
$HIPCC $FLAGS -c igemm_gen1.hip -o $OUT/dbg/igemm_gen1.o &
for f in igemm_bf16 igemm_f32 igemm_i8 igemm_ws; do $HIPCC $FLAGS -c $f.hip -o $OUT/dbg/$f.o & done
$HIPCC $FLAGS -c igemm_bwdn.hip -o $OUT/dbg/igemm_bwdn.o &
$HIPCC $FLAGS -c rowsort.hip -o $OUT/dbg/rowsort.o &
$HIPCC $FLAGS -c norm.hip -o $OUT/dbg/norm.o &
$HIPCC $FLAGS -x hip -c common.cpp -o $OUT/dbg/common.o &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libspconv_amd_dbg.so $OUT/dbg/rulebook.o $OUT/dbg/igemm.o $OUT/dbg/pool.o $OUT/dbg/igemm_gen1.o $OUT/dbg/igemm_bf16.o $OUT/dbg/igemm_f32.o $OUT/dbg/igemm_i8.o $OUT/dbg/igemm_ws.o $OUT/dbg/igemm_bwdn.o $OUT/dbg/rowsort.o $OUT/dbg/norm.o $OUT/dbg/common.o
echo built $OUT/libspconv_amd_dbg.so
