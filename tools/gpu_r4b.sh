#!/bin/bash
# round 4, GPU pass b: parity suite, driver-style bench lines (timing-overhead fit), rulebook device times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4b_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4b_pytest.txt
tail -5 gpurun_out/r4b_pytest.txt
for k in 20 40 80 160; do
  timeout 300 python bench.py --steps $k --warmup 5 --no-also --no-cpu-baseline > gpurun_out/r4b_bench_k$k.json 2> gpurun_out/r4b_bench_k$k.err; echo "bench k$k rc $?"
done
HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/r4b_bench_k20_poll.json 2> gpurun_out/r4b_bench_k20_poll.err; echo "bench poll rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4b_bench_driver.json 2> gpurun_out/r4b_bench_driver.err; echo "bench driver rc $?"
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r4b_bench_long.json 2> gpurun_out/r4b_bench_long.err; echo "bench long rc $?"
RB_ONLY_SUBM=1 timeout 600 python tools/rulebook_bench.py > gpurun_out/r4b_rulebook.json 2> gpurun_out/r4b_rulebook.err; echo "rulebook rc $?"
python - <<'PY'
import json
for f in ("r4b_bench_k20", "r4b_bench_k40", "r4b_bench_k80", "r4b_bench_k160", "r4b_bench_k20_poll", "r4b_bench_driver", "r4b_bench_long"):
    try:
        r = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, round(r["value"] / 1e9, 4), r["ms_per_step"], {k: v["ms"] for k, v in r.get("kernels", {}).items()}, r.get("rulebook_device_ms"), r.get("rows_layout_device_ms"), r.get("eager_device_ms_per_step"))
        if "also" in r:
            print({k: (v.get("value"), v.get("ms_per_step"), v.get("error")) for k, v in r["also"].items()})
    except Exception as e:
        print(f, "unreadable", e)
print(open("gpurun_out/r4b_rulebook.json").read()[:3000])
PY
