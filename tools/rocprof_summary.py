#!/usr/bin/env python
"""Per-kernel summary (calls, avg/min/max duration, share) of a rocprofv3 run.

    python tools/rocprof_summary.py <results.db | kernel_trace.csv> [--pmc]

rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; with
`--output-format csv` it writes *_kernel_trace.csv / *_kernel_stats.csv.  Both are accepted.
With --pmc the counter values of a `--pmc` run are averaged per kernel as well."""
import csv
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("spx::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def from_db(path, pmc):
    db = sqlite3.connect(path)
    rows = db.execute("select name, end - start, dispatch_id from kernels").fetchall()
    stats = defaultdict(list)
    for name, dur, _ in rows:
        stats[short(name)].append(dur)
    counters = defaultdict(lambda: defaultdict(list))
    if pmc:
        try:
            q = ("select k.name, p.counter_name, p.value from pmc_events p "
                 "join kernels k on k.dispatch_id = p.dispatch_id")
            for name, cname, val in db.execute(q):
                counters[short(name)][cname].append(val)
        except sqlite3.Error as e:
            print("pmc query failed:", e)
    return stats, counters


def from_csv(path, pmc):
    stats = defaultdict(list)
    counters = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = short(row.get("Kernel_Name") or row.get("Name") or "")
            if "Start_Timestamp" in row:
                stats[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            if pmc and "Counter_Name" in row:
                counters[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return stats, counters


def from_stats_csv(path):
    """*_kernel_stats.csv (rocprofv3 --stats): one aggregated row per kernel."""
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            rows.append((short(row["Name"]), int(row["Calls"]), float(row["TotalDurationNs"]), float(row["MinNs"]),
                         float(row["MaxNs"])))
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share%':>7s}")
    for name, calls, tot, mn, mx in sorted(rows, key=lambda r: -r[2]):
        print(f"{name:70s} {calls:6d} {tot / calls / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:7.1f}")


def main():
    path = sys.argv[1]
    pmc = "--pmc" in sys.argv
    if path.endswith("kernel_stats.csv"):
        return from_stats_csv(path)
    stats, counters = (from_db if path.endswith(".db") else from_csv)(path, pmc)
    total = sum(sum(v) for v in stats.values()) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share%':>7s}")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name:70s} {len(v):6d} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} "
              f"{max(v) / 1e3:9.2f} {100 * sum(v) / total:7.1f}")
    if pmc:
        print()
        for name, cs in counters.items():
            print(name)
            for cname, vals in sorted(cs.items()):
                print(f"    {cname:32s} avg {sum(vals) / len(vals):16.1f}  (n={len(vals)})")


if __name__ == "__main__":
    main()
