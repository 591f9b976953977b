#!/bin/bash
# round 3, run X: whole GPU suite + smoke + default bench line after the rulebook changes
cd "$(dirname "$0")/.."
O=gpurun_out
(time timeout -k 10 1500 python -m pytest tests -q -m gpu -x) > $O/r3x_pytest.txt 2>&1
echo "rc=$?" >> $O/r3x_pytest.txt
tail -6 $O/r3x_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time timeout -k 10 600 python bench.py) > $O/r3x_bench.json 2> $O/r3x_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3x_bench.json') if l.startswith('{')][-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'roof', r['roofline']['frac'], r['roofline']['traffic'], 'eager', r['eager_device_ms_per_step'], 'cpu', r['cpu_baseline']['value'])
for k, v in r.get('also', {}).items():
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'kernels_ms', 'error', 'eager_ms_per_step', 'live_rows_identical_to_eager')}, v.get('roofline', {}).get('frac'))
PY
