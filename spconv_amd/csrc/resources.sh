#!/bin/bash
# Prints VGPR/AGPR/spill/LDS/occupancy per kernel of a .hip file (gfx950).
f=${1:-igemm.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/res_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n; n=$3; next} /VGPRs:|AGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size/{n=n" | "$0} END{print n}' \
 | sed -E 's/_ZN3spx12_GLOBAL__N_1[0-9]+//; s/EvNS0_.*E \|/ |/' | cut -c1-230
rm -f /tmp/res_$$.o
