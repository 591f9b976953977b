#!/bin/bash
# L2 hit rate and fabric-side fetch of the narrow-layer forward kernels (tools/wsl_probe.py, eager launches):
#   bash tools/wsl_pmc.sh        -> gpurun_out/wsl_pmc.txt
export TMPDIR=/tmp WSL_EAGER=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $O/prof_wsl_$tag -o p -- python $R/tools/wsl_probe.py > $O/wsl_pmc_$tag.log 2>&1); echo "$tag rc=$?"
done
python - <<'PY' > $O/wsl_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof_wsl_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'igemm' not in k: continue
        k = k.split('(')[0].replace('void spx::(anonymous namespace)::', '')
        acc[(k, r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), c in sorted(acc.items()):
    print(k, 'grid', g, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, 'n', {n: len(v) for n, v in c.items()})
PY
cat $O/wsl_pmc.txt; find $O -name "*kernel_trace.csv" -delete
