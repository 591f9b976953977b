#!/bin/bash
# Ordered kernel sequence of one captured config-4 step / config-4i pass (rocprofv3 --kernel-trace, one branch: SPCONV_AMD_PREFETCH=0):
#   [ANCHOR=<first kernel of a captured step, default key_count_kernel: the entry sort>] bash tools/cfg4_sequence.sh [tag]   -> gpurun_out/<tag>_cfg4_sequence.txt, <tag>_cfg4i_sequence.txt
cd "$(dirname "$0")/.."
O=gpurun_out
T=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $O
for cfg in 4 4i; do
  (cd /tmp && SPCONV_AMD_PREFETCH=${PREFETCH:-0} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/seq_${T}_cfg$cfg -o bench -- python $R/bench.py --config $cfg --steps 12 --warmup 4 --no-cpu-baseline ${BENCH_EXTRA} > $R/$O/${T}_cfg${cfg}_seq.log 2>&1)
  f=$(find $O/seq_${T}_cfg$cfg -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && { echo "# tree: $(cat BUILD_STAMP 2>/dev/null || echo unknown: run through tools/grun.sh)"; python tools/step_sequence.py "$f" ${ANCHOR:-key_count_kernel}; } > $O/${T}_cfg${cfg}_sequence.txt 2>&1
  find $O/seq_${T}_cfg$cfg -name "*.csv" -delete
done
