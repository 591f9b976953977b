#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4r_sweep.txt
for hr in 64 32 16 64 32; do
SPX_HINT_ROWS=$hr timeout 600 python bench.py --config 5 --no-also --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/r4r_bench.json 2> gpurun_out/r4r_bench.err; echo "bench $hr rc $?"
python - "$hr" <<'PY' | tee -a gpurun_out/r4r_sweep.txt
import json, sys
r = json.loads(open("gpurun_out/r4r_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(r["value"] / 1e9, 4), round(r["ms_per_step"] * 1e3, 2))
PY
done
