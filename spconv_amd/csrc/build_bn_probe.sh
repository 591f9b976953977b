#!/bin/bash
# Measurement build of norm.hip with SPX_BN_PROBE (SPX_BN_PHASES picks which launches of a BatchNorm call are issued, so
# that tools/bn_probe.py can time each alone): lib/libspconv_amd_bnprobe.so, selected with SPX_LIB.  Product build
# (csrc/build.sh) first: the other objects are linked from it.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT/abl
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
$HIPCC $FLAGS -DSPX_BN_PROBE -c norm.hip -o $OUT/abl/norm_probe.o
OBJS=""
for o in $(bash build.sh --list); do
  if [ "$o" = "norm.o" ]; then OBJS="$OBJS $OUT/abl/norm_probe.o"; else OBJS="$OBJS $OUT/$o"; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libspconv_amd_bnprobe.so $OBJS
echo built $OUT/libspconv_amd_bnprobe.so
