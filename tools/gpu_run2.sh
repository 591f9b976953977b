#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
echo "== kbench default"; timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_default.json
echo "== kbench depth2 chunk128"; SPX_GEMM_DEPTH=2 SPX_WGRAD_CHUNK=128 timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_d2_c128.json
echo "== kbench chunk512"; SPX_WGRAD_CHUNK=512 timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_c512.json
echo "== bench graph sort"; timeout 600 python bench.py --sort > gpurun_out/bench_graph_sort.json 2> gpurun_out/bench_graph_sort.err; echo "rc=$?"; cat gpurun_out/bench_graph_sort.json; tail -3 gpurun_out/bench_graph_sort.err
echo "== bench graph nosort"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "rc=$?"; cat gpurun_out/bench_graph.json
R=$GRAFT_REPO_ROOT
echo "== rocprof stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1b -o bench -- python $R/bench.py --sort --steps 50 --warmup 10 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_r1b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
echo "== rocprof pmc1"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_r1b_pmc1 -o bench -- python $R/bench.py --sort --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc1.log 2>&1); echo "rc=$?"
echo "== rocprof pmc2"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/prof_r1b_pmc2 -o bench -- python $R/bench.py --sort --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc2.log 2>&1); echo "rc=$?"
echo "== rocprof pmc3"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_r1b_pmc3 -o bench -- python $R/bench.py --sort --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof_pmc3.log 2>&1); echo "rc=$?"
ls -R gpurun_out | head -60
du -sh gpurun_out
