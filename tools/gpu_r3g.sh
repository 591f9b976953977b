#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'sort', r['config']['mask_sort'], 'eager', r['eager_device_ms_per_step'], 'rulebook', r['rulebook_ms'])
PY
}
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fused_bwd.py tests/test_gpu_rulebook.py tests/test_gpu_norm.py tests/test_gpu_modules.py -q -x > $O/r3g_pytest.txt 2>&1; echo "rc=$?" >> $O/r3g_pytest.txt; tail -4 $O/r3g_pytest.txt
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 800 > $O/r3g_auto.json 2>> $O/r3g.err; show $O/r3g_auto.json
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 800 --sort off > $O/r3g_off.json 2>> $O/r3g.err; show $O/r3g_off.json
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 400 --config 2b > $O/r3g_2b.json 2>> $O/r3g.err; show $O/r3g_2b.json
