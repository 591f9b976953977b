#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 300 python -u tools/hostprof_net.py 3 > $O/r3k_hostprof3.txt 2>&1; grep -v "amdgpu.ids\|is_fx_tracing" $O/r3k_hostprof3.txt | head -50 | cut -c1-150
timeout -k 10 300 python -u tools/hostprof_layer.py > $O/r3k_hostprof_layer.txt 2>&1; grep "eager step\|host time" $O/r3k_hostprof_layer.txt
