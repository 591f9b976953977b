#!/bin/bash
# Builds libspconv_amd.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16"
pids=()
for f in rulebook igemm igemm_bf16 igemm_f32 igemm_i8 igemm_gen1 igemm_ws igemm_bwdn pool rowsort norm; do
  if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common.h -nt $OUT/$f.o ] || [ igemm_defs.h -nt $OUT/$f.o ] || [ igemm_v4.h -nt $OUT/$f.o ] || [ igemm_bwd.h -nt $OUT/$f.o ] || [ ../../include/spconv_amd.h -nt $OUT/$f.o ]; then
    rm -f $OUT/$f.o
    $HIPCC $FLAGS -c $f.hip -o $OUT/$f.o &
    pids+=($!)
  fi
done
if [ ! -f $OUT/common.o ] || [ common.cpp -nt $OUT/common.o ] || [ common.h -nt $OUT/common.o ]; then
  rm -f $OUT/common.o
  $HIPCC $FLAGS -x hip -c common.cpp -o $OUT/common.o &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done   # a failed compile aborts the build (set -e)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libspconv_amd.so $OUT/rulebook.o $OUT/igemm.o $OUT/igemm_bf16.o $OUT/igemm_f32.o $OUT/igemm_i8.o $OUT/igemm_gen1.o $OUT/igemm_ws.o $OUT/igemm_bwdn.o $OUT/pool.o $OUT/rowsort.o $OUT/norm.o $OUT/common.o
echo built $OUT/libspconv_amd.so
