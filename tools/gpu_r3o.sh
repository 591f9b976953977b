#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
export SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so
for m in i8 i8sort; do
timeout -k 10 200 python tools/timeline.py uniform $m 2>&1 | grep -v amdgpu.ids > $O/r3o_timeline_$m.json
python - $O/r3o_timeline_$m.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1]); print(' phases p10/50/90/100', {k: v for k, v in r['phases_us'].items() if 'staged' not in k}); print(' lifetime', r['wg_lifetime_us'], 'tiles', r['tiles'], 'xcd_span', r['xcd_span_us'][:3])
PY
done
