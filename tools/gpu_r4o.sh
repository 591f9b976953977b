#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "0 1 0" "2 2 0" "2 3 0" "3 2 0" "4 2 0" "0 1 100" "2 2 100" "4 1 50" "8 1 26" "0 1 50" "2 2 50" "0 1 0" "2 2 0"; do
set -- $v
SPX_APP_H=$1 SPX_APP_MULT=$2 SPX_WG_ADJ=$3 timeout 600 python bench.py --config 2 --no-also --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/r4o_bench.json 2> gpurun_out/r4o_bench.err; echo "bench $v rc $?"
python - "$v" <<'PY' | tee -a gpurun_out/r4o_sweep.txt
import json, sys
r = json.loads(open("gpurun_out/r4o_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(r["value"] / 1e9, 4), round(r["ms_per_step"] * 1e3, 2), {k: round(v["ms"] * 1e3, 2) for k, v in r.get("kernels", {}).items()})
PY
done
