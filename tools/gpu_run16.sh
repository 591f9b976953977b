#!/bin/bash
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6
for m in 1 0; do echo "== f32 bench MFMA=$m"; SPX_F32_MFMA=$m timeout 300 python bench.py --dtype f32 --no-cpu-baseline --steps 20 --warmup 5 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"; done
