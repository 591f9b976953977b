#!/bin/bash
# round 3 final evidence: full GPU test suite, smoke, default bench, rocprof + PMC sets of configs 2, 2b, 5
cd "$(dirname "$0")/.."
O=gpurun_out
ulimit -c 0
(time timeout -k 10 900 python -m pytest tests -q -m gpu) > $O/r3s_pytest.txt 2>&1
echo "rc=$?" >> $O/r3s_pytest.txt
tail -5 $O/r3s_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_profile.sh r03 > $O/r3s_profile.log 2>&1
bash tools/gpu_profile_cfg.sh r03_2b fixture-f16-c64-n100000 --config 2b 2>&1 | tail -1 | cut -c1-300
bash tools/gpu_profile_cfg.sh r03_5 uniform-i8-c128-n200000 --config 5 2>&1 | tail -1 | cut -c1-300
rm -rf $O/prof_*_sq $O/prof_*_fetch $O/prof_*_write
find $O -name "*kernel_trace.csv" -delete
(time timeout -k 10 600 python bench.py) > $O/r3s_bench.json 2> $O/r3s_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3s_bench.json') if l.startswith('{')][-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'roof', r['roofline']['frac'], r['roofline']['traffic'], 'eager', r['eager_device_ms_per_step'], 'cpu', r['cpu_baseline']['value'])
for k, v in r.get('also', {}).items():
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'kernels_ms', 'error')}, v.get('roofline', {}).get('frac'), v.get('roofline', {}).get('traffic'))
PY
du -sh $O
