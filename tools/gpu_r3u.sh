#!/bin/bash
# round 3, run U: static-shape inference (rulebook without read-back, captured backbone)
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_static.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/r3u_tests.txt
cat $O/r3u_tests.txt
timeout -k 10 600 python bench.py --config 4i --steps 100 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -5 > $O/r3u_bench_4i.json
cat $O/r3u_bench_4i.json
