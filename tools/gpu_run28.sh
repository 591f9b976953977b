#!/bin/bash
export TMPDIR=/tmp
for v in 0 1; do
HIP_FORCE_DEV_KERNARG=$v python bench.py --no-cpu-baseline --steps 240 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('devkernarg=$v', round(d['ms_per_step']*1e3,2), round(d['ms_per_step_one_step_per_replay']*1e3,2), {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()}, d['eager_device_ms_per_step'], d['rulebook_ms'])"
done
HIP_FORCE_DEV_KERNARG=1 python tools/netbench.py lidar 4 2>&1 | tail -1
