#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 0 768; do
(cd /tmp && SPX_WGRAD_G=$g timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g$g -o g -- python $R/tools/gsweep.py fixture > /dev/null 2>&1)
f=$(find gpurun_out/prof_g$g -name "*kernel_stats.csv" | head -1)
echo "G=$g"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad' in r['Name'] or 'igemm' in r['Name']:
        print(f"  {r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.2f}")
PY
done
