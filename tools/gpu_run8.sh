#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for cfg in "2 0" "2 320" "2 512"; do set -- $cfg
  echo "== kbench WGRAD_V=$1 G=$2"; SPX_WGRAD_V=$1 SPX_WGRAD_G=$2 timeout 100 python tools/kbench.py 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print([(r['scene'],r['sort'],r['wgrad_us']) for r in d['rows']])
except Exception as e: print('failed', e)"
done
export KB_SCENES=uniform KB_SORT=0
for cfg in "2 0"; do set -- $cfg
(cd /tmp && SPX_WGRAD_V=$1 SPX_WGRAD_G=$2 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w$1_$2 -o k -- python $R/tools/kbench.py > /dev/null 2>&1)
f=$(find gpurun_out/prof_w$1_$2 -name "*kernel_stats.csv" | head -1); echo "== V=$1 G=$2"; [ -n "$f" ] && grep "wgrad" "$f" | cut -d, -f1-4 | sed 's/spx::(anonymous namespace):://g' | cut -c1-150
done
