#!/bin/bash
# Builds libspconv_amd.so for gfx950 (MI355X).  Cross-compiles without a GPU.
#   build.sh            rebuild what is older than its sources
#   build.sh --force    delete every object and the library first (the ONE list of translation units is below)
#   build.sh --list     print the object files the library is linked from, one per line
set -e
cd "$(dirname "$0")"
OUT=../lib
HIP_UNITS="rulebook igemm igemm_bf16 igemm_f32 igemm_i8 igemm_gen1 igemm_ws igemm_bwdn pool rowsort norm"
CPP_UNITS="common"
OBJS=""
for f in $HIP_UNITS $CPP_UNITS; do OBJS="$OBJS $OUT/$f.o"; done
if [ "$1" = "--list" ]; then
  for o in $OBJS; do echo "$(basename $o)"; done
  exit 0
fi
mkdir -p $OUT
if [ "$1" = "--force" ]; then
  rm -f $OBJS $OUT/libspconv_amd.so
fi
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16"
pids=()
for f in $HIP_UNITS; do
  if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common.h -nt $OUT/$f.o ] || [ igemm_defs.h -nt $OUT/$f.o ] || [ igemm_v4.h -nt $OUT/$f.o ] || [ igemm_bwd.h -nt $OUT/$f.o ] || [ ../../include/spconv_amd.h -nt $OUT/$f.o ]; then
    rm -f $OUT/$f.o
    $HIPCC $FLAGS -c $f.hip -o $OUT/$f.o &
    pids+=($!)
  fi
done
for f in $CPP_UNITS; do
  if [ ! -f $OUT/$f.o ] || [ $f.cpp -nt $OUT/$f.o ] || [ common.h -nt $OUT/$f.o ]; then
    rm -f $OUT/$f.o
    $HIPCC $FLAGS -x hip -c $f.cpp -o $OUT/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done   # a failed compile aborts the build (set -e)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libspconv_amd.so $OBJS
echo built $OUT/libspconv_amd.so
