#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_v5.py -x -q -k "sparse_kernel" > $O/r3b_sp_tests.txt 2>&1
echo "rc=$?" >> $O/r3b_sp_tests.txt
tail -6 $O/r3b_sp_tests.txt
for sc in uniform fixture; do
  timeout -k 10 300 python tools/v5_bench.py --scene $sc 2>&1 | grep -v amdgpu.ids >> $O/r3b_sp_bench.txt
done
timeout -k 10 300 python tools/v5_bench.py --scene uniform --voxels 90000 2>&1 | grep -v amdgpu.ids >> $O/r3b_sp_bench.txt
timeout -k 10 300 python tools/v5_bench.py --scene uniform --voxels 200000 2>&1 | grep -v amdgpu.ids >> $O/r3b_sp_bench.txt
cat $O/r3b_sp_bench.txt | cut -c1-400
