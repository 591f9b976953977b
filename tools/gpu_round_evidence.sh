#!/bin/bash
# round 4 evidence: full GPU test suite, smoke, default bench as the driver runs it, rocprof + PMC sets of configs 2, 2b, 5,
# kernel stats of the config-4 step, rulebook device times
cd "$(dirname "$0")/.."
O=gpurun_out
T=${1:-r04}
export TMPDIR=/tmp
ulimit -c 0
(time timeout -k 10 1500 python -m pytest tests -q -m gpu) > $O/${T}_pytest.txt 2>&1
echo "rc=$?" >> $O/${T}_pytest.txt
tail -5 $O/${T}_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_profile.sh $T > $O/${T}_profile.log 2>&1
bash tools/gpu_profile_cfg.sh ${T}_2b fixture-f16-c64-n100000 --config 2b 2>&1 | tail -1 | cut -c1-300
bash tools/gpu_profile_cfg.sh ${T}_5 uniform-i8-c128-n200000 --config 5 2>&1 | tail -1 | cut -c1-300
rm -rf $O/prof_*_sq $O/prof_*_fetch $O/prof_*_write
find $O -name "*kernel_trace.csv" -delete
(time timeout -k 10 900 python bench.py --steps 20 --warmup 5) > $O/${T}_bench_driver.json 2> $O/${T}_bench_driver.err
echo "bench rc=$?"
(time timeout -k 10 600 python bench.py --no-also) > $O/${T}_bench_long.json 2> $O/${T}_bench_long.err
RB_ONLY_SUBM=1 timeout 600 python tools/rulebook_bench.py > $O/${T}_rulebook.json 2> $O/${T}_rulebook.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_${T}_cfg4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 40 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/${T}_cfg4_rocprof.log 2>&1)
f=$(find $O/prof_${T}_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" > $O/${T}_cfg4_step_kernels.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import json
for name in ("${T}_bench_driver", "${T}_bench_long"):
    try:
        r = json.loads([l for l in open('gpurun_out/%s.json' % name) if l.startswith('{')][-1])
    except Exception as e:
        print(name, 'unreadable', e); continue
    print(name, 'value', r['value'], 'ms', r['ms_per_step'], 'U', r['config']['steps_per_replay'], 'steady', r.get('steady_state'), 'kernels', {k: v['ms'] for k, v in r['kernels'].items()}, 'roof', r['roofline']['frac'], r['roofline']['traffic'], 'eager', r.get('eager_device_ms_per_step_runs'), 'cpu', (r.get('cpu_baseline') or {}).get('value'), 'rulebook', r.get('rulebook_device_ms'), r.get('rows_layout_device_ms'))
    for k, v in r.get('also', {}).items():
        print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'kernels_ms', 'error')}, v.get('roofline', {}).get('frac'), v.get('roofline', {}).get('traffic'))
print(open('gpurun_out/${T}_rulebook.json').read()[:2500])
PY
du -sh $O
