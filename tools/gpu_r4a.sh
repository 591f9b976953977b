#!/bin/bash
# round 4, first GPU pass: parity suite, headline bench (default flags as the driver runs it + a long run), rulebook device times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4a_pytest.txt
tail -5 gpurun_out/r4a_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench_driver.json 2> gpurun_out/r4a_bench_driver.err; echo "bench rc $?"
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4a_bench_long.json 2> gpurun_out/r4a_bench_long.err; echo "bench long rc $?"
timeout 600 python bench.py --no-also --no-cpu-baseline --sort off > gpurun_out/r4a_bench_long_off.json 2> gpurun_out/r4a_bench_long_off.err; echo "bench off rc $?"
RB_ONLY_SUBM=1 timeout 600 python tools/rulebook_bench.py > gpurun_out/r4a_rulebook.json 2> gpurun_out/r4a_rulebook.err; echo "rulebook rc $?"
python - <<'PY'
import json
for f in ("r4a_bench_driver", "r4a_bench_long", "r4a_bench_long_off"):
    try:
        r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, r["value"], r["ms_per_step"], r.get("kernels", {}).get("fwd"), r.get("kernels", {}).get("bwd"), r.get("rulebook_device_ms"), r.get("rows_layout_device_ms"), r["config"].get("rows_layout"))
        if "also" in r:
            print({k: (v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in r["also"].items()})
    except Exception as e:
        print(f, "unreadable", e)
print(open("gpurun_out/r4a_rulebook.json").read()[:3000])
PY
