#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for mb in 2 1; do echo "== kbench MB=$mb"; SPX_GEMM_MB=$mb timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_v4b_mb$mb.json; done
export SPX_LIB=$PWD/spconv_amd/lib/libspconv_amd_dbg.so
python tools/timeline.py uniform centre 2>&1 | tail -1; python tools/timeline.py uniform 2>&1 | tail -1
