#!/bin/bash
export TMPDIR=/tmp
for g in 128 200 240 300 384; do
echo "== G=$g"; SPX_WGRAD_G=$g timeout 300 python bench.py --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
done
