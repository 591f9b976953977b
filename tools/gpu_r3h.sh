#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/gpu_profile.sh r03 > gpurun_out/r3h_profile.log 2>&1
tail -3 gpurun_out/r3h_profile.log | cut -c1-300
bash tools/gpu_profile_cfg.sh r03_2b fixture-f16-c64-n100000 --config 2b 2>&1 | tail -2 | cut -c1-400
bash tools/gpu_profile_cfg.sh r03_5 uniform-i8-c128-n200000 --config 5 2>&1 | tail -2 | cut -c1-400
# keep what travels back small: only the summaries, the json files and the stats csv
rm -rf gpurun_out/prof_*_sq gpurun_out/prof_*_fetch gpurun_out/prof_*_write
find gpurun_out -name "*kernel_trace.csv" -delete
du -sh gpurun_out
# A/B of the weight pipeline depth (WD = 2 default library vs WD = 1 build) and of the fixture's row order
show() { python - "$1" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'value', round(r['value']/1e9, 3), 'ms', round(r['ms_per_step'], 5), 'kernels', {k: v['ms'] for k, v in r['kernels'].items()})
PY
}
for lib in "" spconv_amd/lib/libspconv_amd_wd1.so; do
  for cfg in 2 2b; do
    SPX_LIB=$lib timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 400 --config $cfg > gpurun_out/r3h_ab.json 2>> gpurun_out/r3h.err; echo "lib=$lib cfg=$cfg"; show gpurun_out/r3h_ab.json
  done
done
BENCH_FIXTURE_ORDER=raster timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --steps 400 --config 2b > gpurun_out/r3h_raster.json 2>> gpurun_out/r3h.err; echo raster; show gpurun_out/r3h_raster.json
timeout -k 10 300 python tools/hostprof_layer.py 2>&1 | grep "eager step\|HIP events\|host time" 
