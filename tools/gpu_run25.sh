#!/bin/bash
export TMPDIR=/tmp
for g in 0 64 128 192 256 512 768; do SPX_WGRAD_G=$g python tools/gsweep.py fixture 2>/dev/null | tail -1; done
SPX_BWD_FUSE=0 python tools/gsweep.py fixture 2>/dev/null | tail -1
SPX_BWD_WGRAD_FIRST=0 python tools/gsweep.py fixture 2>/dev/null | tail -1
for g in 0 128 256 512; do SPX_WGRAD_G=$g python tools/gsweep.py lidar 200000 2>/dev/null | tail -1; done
