timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
bash tools/rb_test.sh
timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('rulebook_ms', d['rulebook_ms'], 'us/step', round(d['ms_per_step']*1e3,2))"
