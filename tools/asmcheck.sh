#!/bin/bash
# Compile igemm.hip for gfx950, keep the ISA, print resource usage and the wait/load skeleton of a kernel's main loop.
#   tools/asmcheck.sh <mangled-name-regex> [lines]
cd /root/repo/spconv_amd/csrc || exit 1
mkdir -p /tmp/t
[ -n "$NOCOMPILE" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -Rpass-analysis=kernel-resource-usage -save-temps=obj -c ${SRC:-igemm.hip} -o /tmp/t/asmcheck.o 2> /tmp/t/asmcheck.log
grep -v "remark:" /tmp/t/asmcheck.log | head -20
S=/tmp/t/$(basename ${SRC:-igemm.hip} .hip)-hip-amdgcn-amd-amdhsa-gfx950.s
name=$(grep -o "^_Z[A-Za-z0-9_]*" $S | grep -E "$1" | head -1)
echo "kernel: $name"
grep -A9 "Function Name: $name" /tmp/t/asmcheck.log | grep -o "SGPRs: [0-9]*\|VGPRs: [0-9]*\|AGPRs: [0-9]*\|ScratchSize.*: [0-9]*\|Occupancy.*: [0-9]*\|LDS Size.*: [0-9]*" | tr '\n' ' '; echo
awk -v n="$name:" 'index($0,n)==1{f=1} f{print} f&&/s_endpgm/{exit}' $S > /tmp/t/kernel.s
echo "lines: $(wc -l < /tmp/t/kernel.s)  accvgpr_mov: $(grep -c accvgpr_mov /tmp/t/kernel.s) readfirstlane: $(grep -c readfirstlane /tmp/t/kernel.s) scratch: $(grep -c scratch_ /tmp/t/kernel.s)"
awk '/Loop Header/{f=1} f{print}' /tmp/t/kernel.s | grep -n "s_barrier\|s_waitcnt vmcnt\|ds_write\|buffer_load\|global_load\|s_cbranch\|Loop Header" | head -${2:-40}
