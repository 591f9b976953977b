#!/bin/bash
export TMPDIR=/tmp KB_SCENES=uniform KB_SORT=0
for mb in 0 1 2; do
SPX_GEMM_MB=$mb python tools/kbench.py 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('mb=$mb', d['cfg5_int8'])"
done
