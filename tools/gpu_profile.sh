#!/bin/bash
# Official measurement set of a round: default bench (with cpu_baseline), rocprofv3 kernel stats and the
# PMC passes of the same command (eager launches so that every kernel is a dispatch), traffic summary.
#   bash tools/gpu_profile.sh <tag>      (run on the GPU box through gpurun)
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 50 --warmup 10 --no-graph --no-cpu-baseline --no-also"
echo "== bench default"; timeout 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "rc=$?"; cat gpurun_out/bench_$TAG.json
echo "== rocprof stats"; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o bench -- $CMD > $R/gpurun_out/rocprof_stats.log 2>&1); echo "rc=$?"
echo "== rocprof pmc SQ"; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_${TAG}_sq -o bench -- $CMD > $R/gpurun_out/rocprof_sq.log 2>&1); echo "rc=$?"
echo "== rocprof pmc FETCH"; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_fetch -o bench -- $CMD > $R/gpurun_out/rocprof_fetch.log 2>&1); echo "rc=$?"
echo "== rocprof pmc WRITE"; (cd /tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_${TAG}_write -o bench -- $CMD > $R/gpurun_out/rocprof_write.log 2>&1); echo "rc=$?"
S=gpurun_out/summary_$TAG.txt
{
echo "# rocprofv3 summaries of: $CMD   (tag $TAG)"
echo "# tree: $(cat BUILD_STAMP 2>/dev/null || echo unknown: run through tools/grun.sh)"
echo "## kernel stats (--kernel-trace --stats)"; f=$(find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" | head -1); cat "$f"
echo "## SQ counters, average per dispatch (quad-cycles)"; f=$(find gpurun_out/prof_${TAG}_sq -name "*counter_collection.csv" | head -1); python tools/rocprof_summary.py "$f" --pmc | sed -n '/^$/,$p'
echo "## FETCH_SIZE [KiB, x2 for wide reads on gfx950]"; f1=$(find gpurun_out/prof_${TAG}_fetch -name "*counter_collection.csv" | head -1); python tools/rocprof_summary.py "$f1" --pmc | sed -n '/^$/,$p'
echo "## WRITE_SIZE [KiB], TCC hit/miss"; f2=$(find gpurun_out/prof_${TAG}_write -name "*counter_collection.csv" | head -1); python tools/rocprof_summary.py "$f2" --pmc | sed -n '/^$/,$p'
} > $S 2>&1
python tools/pmc_traffic.py "$f1" "$f2" uniform-f16-c64-n100000 | tee gpurun_out/traffic_fragment_$TAG.json   # ONE key's entry; the keyed file is profiles/traffic.json
cp profiles/traffic.json gpurun_out/traffic.json
tail -5 gpurun_out/bench_$TAG.err
