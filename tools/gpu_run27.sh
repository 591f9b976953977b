#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline --steps 240 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), {k:round(v['ms']*1e3,2) for k,v in d['kernels'].items()})"
