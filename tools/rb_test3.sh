cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp3 -o k -- python $GRAFT_REPO_ROOT/tools/netbench.py lidar 4 2>/dev/null >/dev/null
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pp3/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    n=r['Name'].replace('spx::(anonymous namespace)::','').split('(')[0][:60]
    print(f"{n:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
