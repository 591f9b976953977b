#!/bin/bash
# v4 gather-GEMM: parity tests, then A/B against v3 and tile-size sweep.
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
for cfg in "3 0" "4 2" "4 1"; do set -- $cfg
  echo "== kbench V=$1 MB=$2"; SPX_GEMM_V=$1 SPX_GEMM_MB=$2 timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_v$1_mb$2.json
done
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err; echo "rc=$?"; cat gpurun_out/bench_v4.json; tail -3 gpurun_out/bench_v4.err
