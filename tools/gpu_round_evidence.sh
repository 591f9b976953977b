#!/bin/bash
# Round evidence in one call: full GPU test suite, smoke, the default bench line as the driver runs it, kernel tables of the
# config-4 step / config-4i pass (both output orders), rulebook device times.     bash tools/gpu_round_evidence.sh [tag]
cd "$(dirname "$0")/.."
O=gpurun_out
T=${1:-r05}
export TMPDIR=/tmp
ulimit -c 0
mkdir -p $O
(time timeout -k 10 1500 python -m pytest tests -q -m gpu) > $O/${T}_pytest.txt 2>&1
echo "rc=$?" >> $O/${T}_pytest.txt
tail -4 $O/${T}_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time timeout -k 10 900 python bench.py --steps 20 --warmup 5) > $O/${T}_bench_driver.json 2> $O/${T}_bench_driver.err
echo "bench rc=$?"
bash tools/cfg4_kernels.sh $T > /dev/null 2>&1
SPCONV_AMD_CONV_ORDER=first_seen bash tools/cfg4_kernels.sh ${T}_first_seen > /dev/null 2>&1
timeout 600 python tools/rulebook_bench.py > $O/${T}_rulebook.json 2> $O/${T}_rulebook.err
python - <<PY
import json
try:
    r = json.loads([l for l in open('gpurun_out/${T}_bench_driver.json') if l.startswith('{')][-1])
    print('value', r['value'], 'ms', r['ms_per_step'], 'r4', r.get('round4_protocol', {}).get('value'), 'steady', r.get('steady_state', {}).get('value'),
          'roof', r['roofline']['frac'], r['roofline']['traffic'], 'eager', r.get('eager_device_ms_per_step'), 'cpu', (r.get('cpu_baseline') or {}).get('value'))
    for k, v in r.get('also', {}).items():
        print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'first_seen_order_ms_per_step', 'eager_ms_per_step', 'error')})
except Exception as e:
    print('bench line unreadable', e)
PY
for f in $O/${T}_cfg4_step_kernels.txt $O/${T}_first_seen_cfg4_step_kernels.txt; do echo "== $f"; head -8 $f; done
du -sh $O
