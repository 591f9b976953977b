#!/bin/bash
# Kernel-level breakdown (rocprofv3 --kernel-trace --stats) of tools/netbench.py: bash tools/netprof.sh [scene] [batch]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_net -o net -- python $R/tools/netbench.py ${1:-lidar} ${2:-4} > $R/gpurun_out/netbench_prof.log 2>&1); echo "rc=$?"
tail -1 gpurun_out/netbench_prof.log
f=$(find gpurun_out/prof_net -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', round(tot/1e6,1))
for r in rows[:32]:
    print(f"{r['Name'].replace('spx::(anonymous namespace)::','')[:80]:80s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
PY
