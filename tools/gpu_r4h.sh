#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_layout.py tests/test_gpu_fused_bwd.py tests/test_gpu_int8.py tests/test_gpu_static.py -x -q > gpurun_out/r4h_pytest.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4h_pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err; echo "bench rc $?"
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4h_bench_long.json 2> gpurun_out/r4h_bench_long.err; echo "bench long rc $?"
python - <<'PY'
import json
for f in ("r4h_bench", "r4h_bench_long"):
    r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(r["value"] / 1e9, 4), r["ms_per_step"], r.get("steady_state"), {k: v["ms"] for k, v in r.get("kernels", {}).items()})
PY
