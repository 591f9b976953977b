#!/bin/bash
# round 3, run AA: SubM rulebook experiments (line-local hashing, masks from a table pass)
cd "$(dirname "$0")/.."
O=gpurun_out
for v in "0 0" "3 0" "0 1" "3 1" "2 0"; do set -- $v
echo "== GBITS=$1 MASK_PASS=$2" >> $O/r3aa.txt
RB_ONLY_SUBM=1 SPX_SUBM_GBITS=$1 SPX_SUBM_MASK_PASS=$2 timeout 300 python tools/rulebook_bench.py 2>/dev/null | tail -1 >> $O/r3aa.txt
done
SPX_SUBM_GBITS=3 SPX_SUBM_MASK_PASS=1 timeout 600 python -m pytest tests/test_gpu_rulebook.py -x -q 2>&1 | tail -2 >> $O/r3aa.txt
cat $O/r3aa.txt
