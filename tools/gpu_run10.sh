#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
for mb in 2 1; do echo "== kbench MB=$mb"; SPX_GEMM_MB=$mb timeout 300 python tools/kbench.py 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print([(r['scene'],r['sort'],r['fwd_us'],r['dgrad_us'],r['wgrad_us'],r.get('fwd_centre_only_us')) for r in d['rows']]); print(d['cfg5_int8'])"; done
