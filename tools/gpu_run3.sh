#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
echo "== kbench default"; timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_default.json
echo "== kbench chunk128"; SPX_WGRAD_CHUNK=128 timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_c128.json
echo "== kbench chunk512"; SPX_WGRAD_CHUNK=512 timeout 300 python tools/kbench.py 2>&1 | tail -1 | tee gpurun_out/kbench_c512.json
echo "== bench graph sort"; timeout 600 python bench.py --sort --no-cpu-baseline > gpurun_out/bench_graph_sort.json 2> gpurun_out/bench_graph_sort.err; echo "rc=$?"; cat gpurun_out/bench_graph_sort.json; tail -3 gpurun_out/bench_graph_sort.err
echo "== bench graph nosort"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "rc=$?"; cat gpurun_out/bench_graph.json
R=$GRAFT_REPO_ROOT
echo "== rocprof stats"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1c -o bench -- python $R/bench.py --steps 50 --warmup 10 --no-graph --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1); echo "rc=$?"
f=$(find gpurun_out/prof_r1c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
