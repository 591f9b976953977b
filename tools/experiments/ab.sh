#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (boxes differ by a few percent):
#   bash tools/ab.sh SPX_SUBM_PROBE 3 4
export TMPDIR=/tmp
VAR=$1; shift
for rep in 1 2; do for val in "$@"; do
  echo "$VAR=$val"
  env $VAR=$val python bench.py --no-cpu-baseline --steps 160 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  uniform us/step', round(d['ms_per_step']*1e3,2), 'rulebook_ms', d['rulebook_ms'])"
  env $VAR=$val python bench.py --no-cpu-baseline --steps 160 --scene lidar 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  lidar   us/step', round(d['ms_per_step']*1e3,2), 'rulebook_ms', d['rulebook_ms'])"
  env $VAR=$val python tools/netbench.py lidar 4 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  cfg3', d['cfg3']['ms_fwd_with_rulebooks'], 'cfg4', d['cfg4']['ms_fwd_bwd_with_rulebooks'], 'infer', d['cfg4']['ms_inference'])"
done; done
