#!/bin/bash
# Kernel sequence of the captured config-2b step (rocprofv3 --kernel-trace): what the 8-step graph replay runs, per kernel.
#   bash tools/cfg2b_sequence.sh [tag]   -> gpurun_out/<tag>_cfg2b_sequence.txt
cd "$(dirname "$0")/.."
O=gpurun_out
T=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/seq_${T}_cfg2b -o bench -- python $R/bench.py --config 2b --steps 40 --warmup 10 --no-cpu-baseline --no-also > $R/$O/${T}_cfg2b_seq.log 2>&1)
f=$(find $O/seq_${T}_cfg2b -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && { echo "# tree: $(cat BUILD_STAMP 2>/dev/null || echo unknown)"; python tools/step_sequence.py "$f" igemm_ws_kernel; } > $O/${T}_cfg2b_sequence.txt 2>&1
python - "$f" <<'PY' >> $O/${T}_cfg2b_sequence.txt
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the LAST 320 kernels of interest = the timed graph replays (3 kernels per step)
names = ("igemm_ws_kernel", "igemm_bwd_kernel", "wgrad_reduce2_kernel")
sel = [(s, e, next(n for n in names if n in k)) for s, e, k in rows if any(n in k for n in names)]
tail = sel[-120:]
agg = collections.defaultdict(list)
for s, e, n in tail:
    agg[n].append((e - s) / 1e3)
print("# last 40 steps (inside the timed replays): per-kernel duration us, mean / min / max")
for n, v in agg.items():
    print(f"#   {n:24s} {sum(v)/len(v):7.2f} {min(v):7.2f} {max(v):7.2f}   n={len(v)}")
span = (tail[-1][1] - tail[0][0]) / 1e3
print(f"#   span of those {len(tail)//3} steps {span:.1f} us -> {span/(len(tail)//3):.2f} us per step; sum of kernel durations {sum(sum(v) for v in agg.values())/(len(tail)//3):.2f} us per step")
PY
find $O/seq_${T}_cfg2b -name "*.csv" -delete
tail -8 $O/${T}_cfg2b_sequence.txt; grep -o '"ms_per_step": [0-9.]*' $O/${T}_cfg2b_seq.log | head -1; grep -o '"kernels": {[^}]*}[^}]*}[^}]*}[^}]*}' $O/${T}_cfg2b_seq.log | head -1 | cut -c1-400
