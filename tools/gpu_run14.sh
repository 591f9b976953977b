#!/bin/bash
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for sc in uniform lidar; do
echo "== bench $sc"; timeout 300 python bench.py --no-cpu-baseline --scene $sc 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
done
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o k -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > /dev/null 2>&1); grep "reduce2\|igemm_bwd\|igemm_v4" $(find /tmp/pp -name "*kernel_stats.csv") | cut -d, -f2-4
