#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
(time timeout -k 10 1500 python -m pytest tests -q -m gpu -x) > $O/r3d_pytest.txt 2>&1
echo "rc=$?" >> $O/r3d_pytest.txt
tail -12 $O/r3d_pytest.txt
(time timeout -k 10 600 python bench.py) > $O/r3d_bench.json 2> $O/r3d_bench.err
echo "bench rc=$?"
tail -3 $O/r3d_bench.err
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r3d_bench.json') if l.startswith('{')][-1])
print('value', r['value'], 'ms', r['ms_per_step'], 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'roof', r['roofline']['frac'])
for k, v in r.get('also', {}).items():
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'kernels_ms', 'wall_s', 'error')}, v.get('roofline', {}).get('frac'))
PY
