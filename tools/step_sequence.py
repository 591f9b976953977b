#!/usr/bin/env python
"""Ordered kernel sequence of ONE step from a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous
kernel's end, name.  The step is the last complete period of the trace (the kernel sequence between two occurrences of
the first kernel of the most frequent period).

    python tools/step_sequence.py <kernel_trace.csv> [anchor-kernel-substring] > sequence.txt
"""
import csv
import sys
from collections import Counter


def short(name):
    name = name.replace("spx::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:64]


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "subm_insert_kernel"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r.get("Kernel_Name") or r.get("Name") or ""),
                         r.get("Queue_Id", ""), r.get("Stream_Id", "")))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < 3:
        print("anchor kernel not found often enough:", anchor, len(idx))
        return
    # period = between the last two anchors whose distance is the most common one
    # the LAST complete period of the trace (bench.py times the captured step last); `--common`: the most frequent one
    if "--common" in sys.argv:
        d = Counter(b - a for a, b in zip(idx[:-1], idx[1:]))
        per = d.most_common(1)[0][0]
        a = [a for a, b in zip(idx[:-1], idx[1:]) if b - a == per][-1]
    else:
        a, per = idx[-2], idx[-1] - idx[-2]
    # walk back from the anchor to the start of the step: the anchor is not necessarily the first kernel; take the
    # window [a - lead, a - lead + per) where lead = kernels between the previous step's last big gap and the anchor
    seq = rows[a:a + per]
    t0 = seq[0][0]
    prev_end = t0
    tot = 0
    print(f"# {per} kernels per step, anchor '{anchor}', step span {(seq[-1][1] - t0) / 1e3:.1f} us")
    print(f"{'#':>4s} {'t_us':>9s} {'dur_us':>8s} {'gap_us':>7s}  q  kernel")
    for i, (s, e, name, q, st) in enumerate(seq):
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:7.2f} {q:>2s}  {name}")
        prev_end = max(prev_end, e)
        tot += e - s
    print(f"# sum of kernel durations {tot / 1e3:.1f} us")


if __name__ == "__main__":
    main()
