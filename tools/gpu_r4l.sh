#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "0 1" "4 1" "8 1" "0 2" "0 4" "2 2"; do
set -- $v
SPX_APP_H=$1 SPX_APP_MULT=$2 timeout 600 python bench.py --config 2 --no-also --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/r4l_bench_$1_$2.json 2> gpurun_out/r4l_bench_$1_$2.err; echo "bench H=$1 MULT=$2 rc $?"
done
python - <<'PY'
import json
for c in ("0_1", "4_1", "8_1", "0_2", "0_4", "2_2"):
    r = json.loads(open(f"gpurun_out/r4l_bench_{c}.json").read().strip().splitlines()[-1])
    print(c, round(r["value"] / 1e9, 4), r["ms_per_step"], {k: v["ms"] for k, v in r.get("kernels", {}).items()})
PY
for h in 0 8; do
for mode in sort bwd; do
SPX_APP_H=$h SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so timeout 300 python tools/timeline.py uniform $mode > gpurun_out/r4l_tl_${mode}_$h.json 2> gpurun_out/r4l_tl_${mode}_$h.err; echo "timeline $mode H=$h rc $?"
cat gpurun_out/r4l_tl_${mode}_$h.json
done
done
