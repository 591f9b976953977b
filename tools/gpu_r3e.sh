#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
(time timeout -k 10 1500 python -m pytest tests -q -m gpu) > $O/r3e_pytest.txt 2>&1
echo "rc=$?" >> $O/r3e_pytest.txt
tail -8 $O/r3e_pytest.txt
for extra in "" "--sort"; do
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 800 $extra > $O/r3e_bench$extra.json 2> $O/r3e_bench.err
python - "$O/r3e_bench$extra.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'sort', r['config']['mask_sort'], 'sort_ms', r.get('mask_sort_device_ms'))
PY
done
