#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'sort', r['config']['mask_sort'])
PY
}
SPX_GEMM_MB=1 timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 800 --sort > $O/r3f_sort_mb1.json 2>> $O/r3f.err; show $O/r3f_sort_mb1.json
SPX_GEMM_MB=1 timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 800 > $O/r3f_mb1.json 2>> $O/r3f.err; show $O/r3f_mb1.json
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 400 --sort --config 2b > $O/r3f_2b_sort.json 2>> $O/r3f.err; show $O/r3f_2b_sort.json
timeout -k 10 300 python tools/hostprof_layer.py > $O/r3f_hostprof_layer.txt 2>&1
head -60 $O/r3f_hostprof_layer.txt | cut -c1-160
