cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('rulebook_ms', d['rulebook_ms'])"
grep -v "igemm\|wgrad" $(find /tmp/pp -name "*kernel_stats.csv") | cut -d, -f1-4 | sed 's/spx::(anonymous namespace):://g' | cut -c1-120
