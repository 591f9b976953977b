#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for cfg in "4 2" "4 1"; do set -- $cfg
  echo "== kbench V=$1 MB=$2"; SPX_GEMM_V=$1 SPX_GEMM_MB=$2 timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_v$1_mb$2.json
done
echo "== rocprof pmc1"; (cd /tmp && SPX_GEMM_MB=2 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_r3_pmc1 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc1.log 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_r3_pmc1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" --pmc | grep -A9 "igemm_v4\|wgrad_mfma"
