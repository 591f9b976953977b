#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 900 python -X faulthandler -m pytest tests/test_gpu_conv.py tests/test_gpu_fused_bwd.py tests/test_gpu_modules.py tests/test_gpu_v5.py -q -x > $O/r3j_pytest.txt 2>&1; echo "rc=$?" >> $O/r3j_pytest.txt; tail -25 $O/r3j_pytest.txt | cut -c1-200
timeout -k 10 300 python -X faulthandler tools/hostprof_layer.py > $O/r3j_hostprof.txt 2>&1; grep -v "^ \|^$\|ncalls\|----" $O/r3j_hostprof.txt | tail -30 | cut -c1-200
