#!/bin/bash
export TMPDIR=/tmp
for ov in 0 1; do
SPCONV_AMD_BWD_OVERLAP=$ov python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('overlap=$ov', round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
done
