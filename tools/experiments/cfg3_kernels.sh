export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg3 -o bench -- python $R/bench.py --config 3 --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/cfg3_rocprof.log 2>&1)
f=$(find $R/gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | head -1)
python $R/tools/rocprof_summary.py "$f" | head -40
find $R/gpurun_out -name "*kernel_trace.csv" -delete
