#!/bin/bash
# Per-kernel table (rocprofv3 --kernel-trace --stats) of the config-4 training step and the config-4i inference pass:
#   bash tools/cfg4_kernels.sh [tag]      -> gpurun_out/<tag>_cfg4_step_kernels.txt, <tag>_cfg4i_kernels.txt
cd "$(dirname "$0")/.."
O=gpurun_out
T=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $O
for cfg in 4 4i; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_${T}_cfg$cfg -o bench -- python $R/bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline > $R/$O/${T}_cfg${cfg}_rocprof.log 2>&1)
  f=$(find $O/prof_${T}_cfg$cfg -name "*kernel_stats.csv" | head -1)
  out=$O/${T}_cfg${cfg}_step_kernels.txt
  [ -n "$f" ] && { echo "# tree: $(cat BUILD_STAMP 2>/dev/null || echo unknown: run through tools/grun.sh)   config $cfg, SPCONV_AMD_PREFETCH=${SPCONV_AMD_PREFETCH:-auto}"; python tools/rocprof_summary.py "$f"; } > $out 2>&1
  grep -o '"ms_per_step": [0-9.]*' $O/${T}_cfg${cfg}_rocprof.log | head -1
  find $O -name "*kernel_trace.csv" -delete
done
