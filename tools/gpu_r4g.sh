#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_layout.py tests/test_gpu_int8.py tests/test_gpu_quantization.py tests/test_gpu_bwd_rows.py -x -q > gpurun_out/r4g_pytest.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4g_pytest.txt
timeout 600 python bench.py --config 5 --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/r4g_bench5.json 2> gpurun_out/r4g_bench5.err; echo "bench rc $?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4g_bench5.json").read().strip().splitlines()[-1])
print(round(r["value"] / 1e9, 4), r["ms_per_step"], r["roofline"]["frac"])
PY
