#!/bin/bash
# First GPU session: smoke, parity tests, bench (graph / eager / lidar), rocprof kernel stats.
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== rocminfo" ; (rocminfo | grep -E "gfx|Compute Unit" | head -4) 2>&1
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
echo "== bench graph"; timeout 600 python bench.py > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "rc=$?"; cat gpurun_out/bench_graph.json; tail -3 gpurun_out/bench_graph.err
echo "== bench eager"; timeout 600 python bench.py --no-graph --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; echo "rc=$?"; cat gpurun_out/bench_eager.json; tail -3 gpurun_out/bench_eager.err
echo "== bench lidar"; timeout 600 python bench.py --scene lidar --no-cpu-baseline > gpurun_out/bench_lidar.json 2> gpurun_out/bench_lidar.err; echo "rc=$?"; cat gpurun_out/bench_lidar.json; tail -3 gpurun_out/bench_lidar.err
echo "== bench lidar sort"; timeout 600 python bench.py --scene lidar --sort --no-cpu-baseline > gpurun_out/bench_lidar_sort.json 2> gpurun_out/bench_lidar_sort.err; echo "rc=$?"; cat gpurun_out/bench_lidar_sort.json; tail -3 gpurun_out/bench_lidar_sort.err
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-graph --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1); echo "rocprof rc=$?"
find gpurun_out/prof_r1 -name "*stats*" | head; f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
