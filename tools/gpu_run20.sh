#!/bin/bash
# kernel-level breakdown of the cfg 4 backbone step (lidar-like scenes, batch 4)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_net -o net -- python $R/tools/netbench.py lidar 4 > $R/gpurun_out/netbench_prof.log 2>&1); echo "rc=$?"
tail -2 gpurun_out/netbench_prof.log
f=$(find gpurun_out/prof_net -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:45]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
PY
