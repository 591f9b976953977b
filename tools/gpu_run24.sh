#!/bin/bash
export TMPDIR=/tmp
for g in 0 256 269 300 340 403 540; do
SPX_WGRAD_G=$g python bench.py --no-cpu-baseline --steps 240 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('G=$g', round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
done
