#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_int8.py tests/test_gpu_quantization.py -q -x > $O/r3n_pytest.txt 2>&1; echo "rc=$?" >> $O/r3n_pytest.txt; tail -3 $O/r3n_pytest.txt
for srt in auto off; do
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline --steps 400 --sort $srt > $O/r3n_int8_$srt.json 2>> $O/r3n.err
python - $O/r3n_int8_$srt.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'frac', r['roofline']['frac'], 'sort', r['config'].get('mask_sort'))
PY
done
