#!/bin/bash
# round 3, run V: kernel breakdown of the inference pass (config 4i), eager launches so that every kernel is a dispatch
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_4i -o bench -- python $R/bench.py --config 4i --steps 40 --warmup 5 > $O/rocprof_4i.log 2>&1); echo "rc=$?"
f=$(find $O/prof_4i -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220 > $O/r3v_4i_kernel_stats.txt
cat $O/r3v_4i_kernel_stats.txt
rm -rf $O/prof_4i
