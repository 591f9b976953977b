#!/bin/bash
# rocprof kernel stats of tools/level_probe.py (narrow-layer backward)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lp -o lp -- python $R/tools/level_probe.py > $O/rocprof_lp.log 2>&1); echo "rc=$?"
f=$(find $O/prof_lp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'].replace('spx::(anonymous namespace)::','')[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
rm -rf $O/prof_lp
