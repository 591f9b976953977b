#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_layout.py tests/test_gpu_fused_bwd.py tests/test_gpu_int8.py tests/test_gpu_modules.py tests/test_gpu_bwd_rows.py tests/test_gpu_static.py tests/test_gpu_dist.py -x -q > gpurun_out/r4p_pytest.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4p_pytest.txt
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4p_bench_driver.json 2> gpurun_out/r4p_bench_driver.err; echo "bench driver rc $?"
for cfg in 2 2b 5; do
timeout 600 python bench.py --config $cfg --no-also --no-cpu-baseline --steps 2000 --warmup 50 > gpurun_out/r4p_bench_$cfg.json 2> gpurun_out/r4p_bench_$cfg.err; echo "bench $cfg rc $?"
done
python - <<'PY'
import json
for c in ("driver", "2", "2b", "5"):
    r = json.loads(open(f"gpurun_out/r4p_bench_{c}.json").read().strip().splitlines()[-1])
    print(c, round(r["value"] / 1e9, 4), round(r["ms_per_step"] * 1e3, 2), {k: round(v["ms"] * 1e3, 2) for k, v in r.get("kernels", {}).items()}, r["roofline"]["frac"], r.get("steady_state"))
PY
