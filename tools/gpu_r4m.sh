#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_layout.py tests/test_gpu_fused_bwd.py tests/test_gpu_int8.py tests/test_gpu_modules.py -x -q > gpurun_out/r4m_pytest.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4m_pytest.txt
for v in "0 1" "8 1" "2 2"; do
set -- $v
SPX_APP_H=$1 SPX_APP_MULT=$2 timeout 600 python bench.py --config 2 --no-also --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/r4m_bench_$1_$2.json 2> gpurun_out/r4m_bench_$1_$2.err; echo "bench H=$1 MULT=$2 rc $?"
done
for cfg in 2b 5; do
timeout 600 python bench.py --config $cfg --no-also --no-cpu-baseline --steps 400 --warmup 50 > gpurun_out/r4m_bench_$cfg.json 2> gpurun_out/r4m_bench_$cfg.err; echo "bench $cfg rc $?"
done
python - <<'PY'
import json
for c in ("0_1", "8_1", "2_2", "2b", "5"):
    r = json.loads(open(f"gpurun_out/r4m_bench_{c}.json").read().strip().splitlines()[-1])
    print(c, round(r["value"] / 1e9, 4), r["ms_per_step"], {k: v["ms"] for k, v in r.get("kernels", {}).items()})
PY
for mode in sort bwd; do
SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so timeout 300 python tools/timeline.py uniform $mode > gpurun_out/r4m_tl_${mode}.json 2> gpurun_out/r4m_tl_${mode}.err; echo "timeline $mode rc $?"
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4m_tl_sort.json").read())
print({k: r[k] for k in ("phases_us", "appendix_lifetime_us", "wg_lifetime_us")})
print(open("gpurun_out/r4m_tl_bwd.json").read())
PY
