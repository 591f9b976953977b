#!/bin/bash
# static-shape training step (configs 3 / 4 captured whole) + inference pass (4i)
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_static.py tests/test_gpu_norm.py tests/test_gpu_modules.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > $O/r3y_tests.txt
cat $O/r3y_tests.txt
for c in 3 4 4i; do
timeout -k 10 600 python bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline 2> $O/r3y_bench$c.err | tail -1 > $O/r3y_bench$c.json
python - $c <<'PY'
import json, sys
r = json.loads(open(f'gpurun_out/r3y_bench{sys.argv[1]}.json').read())
print(sys.argv[1], 'value', r['value'], 'ms', r['ms_per_step'], 'eager', r.get('eager_ms_per_step'), r['config']['launch'][:40], r['config'].get('static_shapes'), r.get('live_rows_identical_to_eager'), r['roofline']['frac'])
PY
done
