#!/bin/bash
# Session 2, call 1: default bench (with cpu_baseline), floors, overlap A/B, rocprof stats + PMC passes.
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "== bench default"; timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "rc=$?"; cat gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
echo "== bench no-overlap"; SPCONV_AMD_BWD_OVERLAP=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_nooverlap.json 2> gpurun_out/bench_nooverlap.err; echo "rc=$?"; cat gpurun_out/bench_nooverlap.json
echo "== kbench"; timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_r2.json
echo "== rocprof stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2 -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_r2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
echo "== rocprof pmc1"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_r2_pmc1 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc1.log 2>&1); echo "rc=$?"
echo "== rocprof pmc2"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2_pmc2 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc2.log 2>&1); echo "rc=$?"
echo "== rocprof pmc3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_r2_pmc3 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc3.log 2>&1); echo "rc=$?"
for d in pmc1 pmc2 pmc3; do f=$(find gpurun_out/prof_r2_$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" --pmc | tail -40; done
du -sh gpurun_out
