#!/bin/bash
export TMPDIR=/tmp
for cfg in "0 384" "1 384" "1 242" "0 242"; do set -- $cfg
echo "== wgrad_first=$1 G=$2"; SPX_BWD_WGRAD_FIRST=$1 SPX_WGRAD_G=$2 timeout 300 python bench.py --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
done
