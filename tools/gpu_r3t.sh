#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
export SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so
show() { python - "$1" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'tiles', r['tiles'], 'lifetime', r['wg_lifetime_us'], 'packing', r['packing'], 'loop', r['phases_us'].get('prologue_done->loop_done'))
PY
}
TL_N=400000 TL_C=32 TL_SLOTS=2048 timeout -k 10 200 python tools/timeline.py lidar 2>&1 | grep -v amdgpu.ids > $O/r3t_lidar400k_c32.json; show $O/r3t_lidar400k_c32.json
TL_N=400000 TL_C=64 TL_SLOTS=1024 timeout -k 10 200 python tools/timeline.py lidar 2>&1 | grep -v amdgpu.ids > $O/r3t_lidar400k_c64.json; show $O/r3t_lidar400k_c64.json
TL_N=1000000 TL_C=64 TL_SLOTS=1024 timeout -k 10 200 python tools/timeline.py lidar 2>&1 | grep -v amdgpu.ids > $O/r3t_lidar1m_c64.json; show $O/r3t_lidar1m_c64.json
