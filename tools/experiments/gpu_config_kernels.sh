#!/bin/bash
# kernel stats of config 3 (eager + captured static steps)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python $R/bench.py --config ${CFG:-3} --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline > $O/rocprof_c3.log 2>&1); echo "rc=$?"
f=$(find $O/prof_c3 -name "*kernel_stats.csv" | head -1)
python - "$f" > $O/r3z_cfg${CFG:-3}_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total device ms {tot/1e6:.1f}")
for r in rows[:32]:
    print(f"{r['Name'].replace('spx::(anonymous namespace)::','')[:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {float(r['Percentage']):5.1f}")
PY
cat $O/r3z_cfg${CFG:-3}_kernels.txt
rm -rf $O/prof_c3
