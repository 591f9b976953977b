#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
lscpu | grep -i "numa\|model name\|^CPU(s)" | head -8
for a in none none node:0 node:1 none node:0 node:1 0-7; do
timeout 120 python tools/eager_probe.py $a 2>&1 | tail -1
done | tee gpurun_out/r4s_eager.txt
