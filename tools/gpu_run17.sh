#!/bin/bash
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6
python tools/netbench.py lidar 4 2>&1 | tail -1; python tools/netbench.py lidar 1 2>&1 | tail -1
bash tools/rb_test3.sh 2>&1 | head -14
