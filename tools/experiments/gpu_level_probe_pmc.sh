#!/bin/bash
# PMC passes (L2 requests / hits, HBM fetch) of the narrow-layer backward kernels on the real backbone levels
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ulimit -c 0
for pass in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
tag=$(echo $pass | cut -d' ' -f1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $O/prof_lpc_$tag -o lp -- python $R/tools/level_probe.py > $O/rocprof_lpc.log 2>&1); echo "rc=$?"
f=$(find $O/prof_lpc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" >> $O/r3ac_bwd_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("spx::(anonymous namespace)::", "")
    if not any(t in n for t in ("bwdn_kernel", "igemm_bwd_kernel", "wgrad_tr_kernel", "bwdn_reduce")): continue
    acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in sorted(acc.items()):
    print(n, {k: round(sum(v) / len(v)) for k, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
rm -rf $O/prof_lpc_$tag
done
cat $O/r3ac_bwd_pmc.txt
