#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
export SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so
for m in "" sort; do
timeout -k 10 200 python tools/timeline_bwd.py uniform $m 2>&1 | grep -v amdgpu.ids > $O/r3q_timeline_bwd_$m.json
python - $O/r3q_timeline_bwd_$m.json <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'call_us', r['call_us_events'], 'span', r['kernel_span_per_domain'][:4])
print(' wgrad', {k: v for k, v in r['wgrad'].items()})
print(' dgrad', {k: v for k, v in r['dgrad'].items()})
PY
done
