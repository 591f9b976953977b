#!/usr/bin/env python
"""Turns the rocprofv3 PMC passes of `python bench.py` into profiles/traffic.json.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <key>

HBM bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 (gfx950 reports half of a wide coalesced read,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE [KiB] x 1024, averaged over the dispatches of each
kernel; kernels are grouped as bench.py groups them (fwd = the forward gather-GEMM, bwd = fused
dgrad/wgrad launch + wgrad second stage)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    """kernel -> dispatch_id -> counter value"""
    vals = defaultdict(dict)
    args = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            vals[row["Kernel_Name"]][row["Dispatch_Id"]] = float(row["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in vals.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    key = sys.argv[3]
    # the forward and the dgrad kernels are two instantiations of igemm_v4_kernel (last template
    # argument false / true); the backward launch is igemm_bwd_kernel
    groups = {"fwd": [], "bwd": []}
    for name in set(fetch) | set(write):
        b = fetch.get(name, 0.0) * 1024 * 2 + write.get(name, 0.0) * 1024
        m = re.search(r"(igemm_v4_kernel|igemm_ws_kernel|igemm_i8_sparse_kernel|igemm_bwd_kernel|wgrad_reduce2_kernel)<([^>]*)>", name)
        if not m:
            continue
        short = f"{m.group(1)}<{m.group(2)}>"
        targs = [a.strip() for a in m.group(2).split(",")]      # <COUT, MB, DT, BT, NKS>
        if m.group(1) == "igemm_v4_kernel" and len(targs) > 3 and targs[3] == "false":
            groups["fwd"].append((short, b))
        elif m.group(1) == "igemm_ws_kernel" and targs[-1] == "false":      # <NW, G, D, BF16, BT>: the forward launch
            groups["fwd"].append((short, b))
        elif m.group(1) == "igemm_i8_sparse_kernel":                        # int8 forward, sparse class (appendix + streaming tiles)
            groups["fwd"].append((short, b))
        elif m.group(1) in ("igemm_bwd_kernel", "wgrad_reduce2_kernel"):
            groups["bwd"].append((short, b))
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                            "traffic.json")
    try:
        with open(out_path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    data[key] = {g: {"hbm_bytes_per_launch": int(sum(b for _, b in ks)),
                     "kernels": {n: int(b) for n, b in ks}} for g, ks in groups.items()}
    with open(out_path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print(json.dumps(data[key]))


if __name__ == "__main__":
    main()
