#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4q_sweep.txt
for cfg in 2b 2 5; do
for lib in libspconv_amd.so libspconv_amd_mf.so; do
SPX_LIB=$PWD/spconv_amd/lib/$lib timeout 600 python bench.py --config $cfg --no-also --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/r4q_bench.json 2> gpurun_out/r4q_bench.err; echo "bench $cfg $lib rc $?"
python - "$cfg $lib" <<'PY' | tee -a gpurun_out/r4q_sweep.txt
import json, sys
r = json.loads(open("gpurun_out/r4q_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(r["value"] / 1e9, 4), round(r["ms_per_step"] * 1e3, 2), {k: round(v["ms"] * 1e3, 2) for k, v in r.get("kernels", {}).items()})
PY
done
done
