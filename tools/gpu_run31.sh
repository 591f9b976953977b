#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py tests/test_gpu_int8.py -x -q -m gpu 2>&1 | tail -2
python tools/packpot.py 2>/dev/null | tail -1
python tools/netbench.py lidar 4 2>&1 | tail -1
