#!/bin/bash
export TMPDIR=/tmp
for mp in 0 1; do
echo "mask_pass=$mp"
SPX_SUBM_MASK_PASS=$mp python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' uniform rulebook_ms', d['rulebook_ms'])"
SPX_SUBM_MASK_PASS=$mp python bench.py --no-cpu-baseline --steps 100 --scene lidar 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(' lidar rulebook_ms', d['rulebook_ms'], round(d['ms_per_step']*1e3,2))"
SPX_SUBM_MASK_PASS=$mp python tools/netbench.py lidar 4 2>&1 | tail -1
done
timeout 600 python -m pytest tests/test_gpu_rulebook.py tests/test_gpu_modules.py -q -m gpu 2>&1 | tail -2
