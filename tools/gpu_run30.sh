#!/bin/bash
export TMPDIR=/tmp KB_INT8=0 KB_SORT=0 KB_CHANNELS=128
for n in 50000 100000 200000 400000; do for mb in 1 2; do
KB_VOXELS=$n SPX_GEMM_MB=$mb python tools/kbench.py 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C=128 n=$n mb=$mb', [(r['scene'],r['fwd_us'],r['dgrad_us']) for r in d['rows']])"; done; done
