#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_int8.py tests/test_gpu_quantization.py tests/test_gpu_conv.py tests/test_gpu_fused_bwd.py tests/test_gpu_modules.py -q > $O/r3p_pytest.txt 2>&1; echo "rc=$?" >> $O/r3p_pytest.txt; tail -3 $O/r3p_pytest.txt
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline --steps 400 > $O/r3p_int8.json 2>> $O/r3p.err
python - $O/r3p_int8.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'frac', r['roofline']['frac'], 'sort', r['config'].get('mask_sort'))
PY
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 800 > $O/r3p_cfg2.json 2>> $O/r3p.err
python - $O/r3p_cfg2.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'kernels', {k: v['ms'] for k, v in r['kernels'].items()})
PY
