cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline 2>/dev/null >/dev/null
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].replace('spx::(anonymous namespace)::','').split('(')[0]
    if 'igemm' in n or 'wgrad' in n: continue
    print(f"{n:40s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
