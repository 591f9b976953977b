#!/bin/bash
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for sc in uniform lidar; do
echo "== bench $sc"; timeout 300 python bench.py --no-cpu-baseline --scene $sc 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
done
KB_SCENES=uniform KB_SORT=0 timeout 200 python tools/kbench.py 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['rows'], d['cfg5_int8'])"
export SPX_TICK_MHZ=2100 SPX_LIB=$PWD/spconv_amd/lib/libspconv_amd_dbg.so
for m in centre full; do python tools/timeline.py uniform $m 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v for k,v in d['phases_us'].items() if 'staged' not in k}); print(d['wg_lifetime_us'])"; done
