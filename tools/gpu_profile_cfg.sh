#!/bin/bash
# rocprofv3 kernel stats + FETCH / WRITE passes of one bench configuration (eager launches: every kernel is a
# dispatch) -> profiles/traffic.json entry <key>.     bash tools/gpu_profile_cfg.sh <tag> <key> <bench args...>
TAG=$1; KEY=$2; shift 2
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 40 --warmup 10 --no-graph --no-cpu-baseline --no-also $@"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o bench -- $CMD > $R/gpurun_out/rocprof_${TAG}_stats.log 2>&1); echo "stats rc=$?"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_fetch -o bench -- $CMD > $R/gpurun_out/rocprof_${TAG}_fetch.log 2>&1); echo "fetch rc=$?"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_write -o bench -- $CMD > $R/gpurun_out/rocprof_${TAG}_write.log 2>&1); echo "write rc=$?"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_${TAG}_sq -o bench -- $CMD > $R/gpurun_out/rocprof_${TAG}_sq.log 2>&1); echo "sq rc=$?"
S=gpurun_out/summary_$TAG.txt
f1=$(find gpurun_out/prof_${TAG}_fetch -name "*counter_collection.csv" | head -1)
f2=$(find gpurun_out/prof_${TAG}_write -name "*counter_collection.csv" | head -1)
{
echo "# rocprofv3 summaries of: $CMD   (tag $TAG)"
echo "# tree: $(cat BUILD_STAMP 2>/dev/null || echo unknown: run through tools/grun.sh)"
echo "## kernel stats (--kernel-trace --stats)"; f=$(find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" | head -1); head -12 "$f"
echo "## SQ counters, average per dispatch"; f=$(find gpurun_out/prof_${TAG}_sq -name "*counter_collection.csv" | head -1); python tools/rocprof_summary.py "$f" --pmc | sed -n '/^$/,$p'
echo "## FETCH_SIZE [KiB, x2 for wide reads on gfx950]"; python tools/rocprof_summary.py "$f1" --pmc | sed -n '/^$/,$p'
echo "## WRITE_SIZE [KiB]"; python tools/rocprof_summary.py "$f2" --pmc | sed -n '/^$/,$p'
} > $S 2>&1
python tools/pmc_traffic.py "$f1" "$f2" $KEY | cut -c1-400
cp profiles/traffic.json gpurun_out/traffic_$TAG.json
