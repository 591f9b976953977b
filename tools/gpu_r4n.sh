#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "0 1" "8 1" "2 2"; do
set -- $v
SPX_APP_H=$1 SPX_APP_MULT=$2 TL_RAW=gpurun_out/r4n_raw_$1_$2.npy SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so timeout 300 python tools/timeline.py uniform bwd > gpurun_out/r4n_tl_$1_$2.json 2> gpurun_out/r4n_tl_$1_$2.err; echo "timeline $v rc $?"
done
