#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" spconv_amd/lib/libspconv_amd_spec0.so; do
  SPX_LIB=$lib timeout 600 python bench.py --no-also --no-cpu-baseline --steps 1000 --warmup 100 > gpurun_out/r4i_$rep.json 2>/dev/null
  python - <<PY
import json
r = json.loads(open("gpurun_out/r4i_$rep.json").read().strip().splitlines()[-1])
print("lib='$lib'", round(r["value"] / 1e9, 4), r["ms_per_step"], {k: v["ms"] for k, v in r.get("kernels", {}).items()})
PY
done
done
