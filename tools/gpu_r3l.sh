#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
(time timeout -k 10 1500 python -m pytest tests -q -m gpu) > $O/r3l_pytest.txt 2>&1
echo "rc=$?" >> $O/r3l_pytest.txt
tail -6 $O/r3l_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(time timeout -k 10 600 python bench.py) > $O/r3l_bench.json 2> $O/r3l_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3l_bench.json') if l.startswith('{')][-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'roof', r['roofline']['frac'], r['roofline']['traffic'], 'eager', r['eager_device_ms_per_step'], 'rulebook', r['rulebook_ms'], r['rulebook_device_ms'], 'cpu', r['cpu_baseline']['value'])
for k, v in r.get('also', {}).items():
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'kernels_ms', 'error')}, v.get('roofline', {}).get('frac'), v.get('roofline', {}).get('traffic'))
PY
