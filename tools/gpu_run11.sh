#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== kbench"; KB_INT8=0 timeout 300 python tools/kbench.py 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print([(r['scene'],r['sort'],r['fwd_us'],r['dgrad_us'],r['wgrad_us'],r.get('fwd_centre_only_us')) for r in d['rows']])"
export SPX_TICK_MHZ=2100 SPX_LIB=$PWD/spconv_amd/lib/libspconv_amd_dbg.so
python tools/timeline.py uniform centre 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['phases_us']); print(d['wg_lifetime_us'])"
