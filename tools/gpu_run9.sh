#!/bin/bash
export TMPDIR=/tmp KB_SCENES=uniform KB_SORT=0
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for cfg in "1 0" "2 256" "2 512"; do set -- $cfg
(cd /tmp && SPX_WGRAD_V=$1 SPX_WGRAD_G=$2 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w$1_$2 -o k -- python $R/tools/kbench.py > /dev/null 2>&1)
f=$(find gpurun_out/prof_w$1_$2 -name "*kernel_stats.csv" | head -1); echo "== V=$1 G=$2"; [ -n "$f" ] && grep "wgrad" "$f" | cut -d, -f1-4,6,7 | sed 's/spx::(anonymous namespace):://g' | cut -c1-150
done
