#!/bin/bash
# round 4, GPU pass c: full parity suite + the driver-style line (one graph of K steps) + int8 with the host hint
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4c_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4c_pytest.txt
tail -5 gpurun_out/r4c_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4c_bench_driver.json 2> gpurun_out/r4c_bench_driver.err; echo "bench driver rc $?"
python - <<'PY'
import json
for f in ("r4c_bench_driver",):
    try:
        r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r["value"] / 1e9, 4), r["ms_per_step"], r["config"]["steps_per_replay"], r.get("steady_state"), {k: v["ms"] for k, v in r.get("kernels", {}).items()}, r.get("rulebook_device_ms"), r.get("eager_device_ms_per_step_runs"))
        if "also" in r:
            print({k: (v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in r["also"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
