#!/bin/bash
# PMC passes over the standalone wgrad (tools/wgrad_probe.py, one width); every pass under its own timeout.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PROBE_PMC=1 PROBE_ONLY_WGRAD=1 PROBE_N=${PROBE_N:-1200000} PROBE_C=${PROBE_C:-32}
i=0
for set in \
 "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
 "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU" ; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set -d /tmp/pmc$i -o p --output-format csv -- python $R/tools/wgrad_probe.py > /tmp/pmc$i.log 2>&1
  echo "pass $i rc=$?"
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "wgrad_tr" in k:
        acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in sorted(acc.items()):
    vals = [sum(v) for v in d.values()]
    print(f"  {c:40s} avg per dispatch {sum(vals)/len(vals):16.1f}  (n={len(vals)})")
PY
done
