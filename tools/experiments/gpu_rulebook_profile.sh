#!/bin/bash
# compact-candidate regular-conv rulebook (SPX_CONV_V=3) -- parity, then device times v2 vs v3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_rulebook.py tests/test_gpu_static.py -x -q 2>&1 | grep -v amdgpu.ids | tail -25 > $O/r3w_tests.txt
cat $O/r3w_tests.txt
timeout -k 10 600 python tools/rulebook_bench.py 2>&1 | grep -v amdgpu.ids | tail -3 > $O/r3w_rulebook_bench.json
cat $O/r3w_rulebook_bench.json
fi
for sc in "lidar 4" "uniform 1"; do set -- $sc
for v in 2 3; do
(cd /tmp && RB_SCENE=$1 RB_BATCH=$2 SPX_CONV_V=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rb -o rb -- python $R/tools/rulebook_loop.py > $O/rocprof_rb.log 2>&1)
f=$(find $O/prof_rb -name "*kernel_stats.csv" | head -1)
echo "== $1 x$2 SPX_CONV_V=$v" >> $O/r3w_rulebook_kernels.txt
python - "$f" >> $O/r3w_rulebook_kernels.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'].replace('spx::(anonymous namespace)::','')[:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
rm -rf $O/prof_rb
done; done
cat $O/r3w_rulebook_kernels.txt
