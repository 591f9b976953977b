#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ns_probe.py > gpurun_out/r4d_ns_probe.txt 2>&1; echo "ns probe rc $?"; cat gpurun_out/r4d_ns_probe.txt | grep -v amdgpu.ids | tail -8
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench.py > gpurun_out/r4d_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4d_pytest.txt
tail -4 gpurun_out/r4d_pytest.txt
