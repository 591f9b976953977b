#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
for f in 1 0; do for sc in uniform lidar; do
echo "== bench fuse=$f $sc"; SPX_BWD_FUSE=$f timeout 300 python bench.py --no-cpu-baseline --scene $sc 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', round(d['value']/1e9,3),'Gvox/s', d['kernels'])"
done; done
