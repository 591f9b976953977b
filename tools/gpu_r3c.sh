#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
export SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so
timeout -k 10 200 python tools/timeline.py uniform 2>&1 | grep -v amdgpu.ids > $O/r3c_timeline_v4.json
timeout -k 10 200 python tools/timeline.py uniform sp 2>&1 | grep -v amdgpu.ids > $O/r3c_timeline_sp.json
cat $O/r3c_timeline_v4.json | cut -c1-1800
echo
cat $O/r3c_timeline_sp.json | cut -c1-1800
