#!/bin/bash
# round 3, GPU call A: LDS-DMA probe, v5 parity, v4-vs-v5 timing
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
./tools/probes/dma_probe > $O/r3a_dma_probe.txt 2>&1
echo "probe rc=$?" >> $O/r3a_dma_probe.txt
timeout -k 10 200 python -m pytest tests/test_gpu_v5.py -x -q -k "test_v5_vs_oracle" > $O/r3a_v5_first.txt 2>&1
echo "rc=$?" >> $O/r3a_v5_first.txt
tail -5 $O/r3a_v5_first.txt
if grep -q "rc=0" $O/r3a_v5_first.txt; then
  timeout -k 10 600 python -m pytest tests/test_gpu_v5.py -q > $O/r3a_v5_tests.txt 2>&1
  echo "rc=$?" >> $O/r3a_v5_tests.txt
  tail -15 $O/r3a_v5_tests.txt
  for sc in uniform fixture; do
    timeout -k 10 300 python tools/v5_bench.py --scene $sc >> $O/r3a_v5_bench.txt 2>&1
    SPX_V5_VARIANT=1 timeout -k 10 300 python tools/v5_bench.py --scene $sc >> $O/r3a_v5_bench.txt 2>&1
    SPX_V5_WGS=256 timeout -k 10 300 python tools/v5_bench.py --scene $sc >> $O/r3a_v5_bench.txt 2>&1
    SPX_V5_WGS=768 timeout -k 10 300 python tools/v5_bench.py --scene $sc >> $O/r3a_v5_bench.txt 2>&1
  done
  grep -v "^\[" $O/r3a_v5_bench.txt | cut -c1-600
fi
cat $O/r3a_dma_probe.txt | head -40
