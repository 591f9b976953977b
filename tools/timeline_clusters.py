#!/usr/bin/env python
"""Per-clock-domain summary of a raw stamp table of tools/timeline.py (TL_RAW=file.npy, `bwd` mode).

    python tools/timeline_clusters.py raw.npy <napp>

s_memtime is not synchronised across the chip: stamps are clustered by their entry value (gaps > 50 us separate
domains; clusters of more than 20 workgroups are merged domains and are left out) and every figure is a
difference inside one cluster.  Prints, per kind of workgroup (wgrad range | appendix |
dgrad main tile), lifetimes over all clusters, and the distribution of the clusters' spans (first entry -> last retire),
split by whether a workgroup of the cluster entered late (> 3 us after the first)."""
import json
import sys

import numpy as np

TICK = 1 / 2200.0


def main():
    a = np.load(sys.argv[1]).astype(np.int64)
    napp = int(sys.argv[2])
    ntiles = int(sys.argv[3]) if len(sys.argv) > 3 else 782
    last = int(np.nonzero(a[:, 0] > 0)[0].max()) + 1
    a = a[:last]
    nw = last - ntiles - napp
    kind = np.zeros(last, int)
    kind[nw:nw + napp] = 1
    kind[nw + napp:] = 2
    order = np.argsort(a[:, 0])
    cuts = np.nonzero(np.diff(a[order, 0]) * TICK > 50)[0]
    groups = [g for g in np.split(order, cuts + 1) if (a[g, 7] > 0).any()]
    life = {0: [], 1: [], 2: []}
    spans_all, spans_late, n_late = [], [], 0
    used = 0
    for g in groups:
        if len(g) > 20:              # several domains whose clocks lie within 50 us of each other: not separable
            continue
        used += 1
        A, k = a[g], kind[g]
        done = A[:, 7] > 0
        t0 = A[:, 0].min()
        ent = (A[:, 0] - t0) * TICK
        ret = (A[:, 7] - t0) * TICK
        for kk in (0, 1, 2):
            life[kk].extend(((A[:, 7] - A[:, 0]) * TICK)[done & (k == kk)].tolist())
        late = done & (ent > 3.0)
        n_late += int(late.sum())
        (spans_late if late.any() else spans_all).append(float(ret[done].max()))
    pct = lambda v, q: [round(float(x), 2) for x in np.percentile(v, q)] if len(v) else None
    print(json.dumps({
        "workgroups": last, "wgrad_ranges": nw, "appendix_workgroups": napp, "dgrad_tiles": ntiles,
        "clock_domains": len(groups), "domains_used": used,
        "lifetime_us_p10_p50_p90_max": {n: pct(life[i], [10, 50, 90, 100]) for i, n in enumerate(("wgrad", "appendix", "dgrad"))},
        "appendix_with_rows": len(life[1]),
        "span_us_domains_where_all_started_at_once_p10_p50_p90": pct(spans_all, [10, 50, 90]),
        "span_us_domains_with_a_late_workgroup_p10_p50_p90": pct(spans_late, [10, 50, 90]),
        "domains_at_once": len(spans_all), "domains_with_late": len(spans_late), "late_workgroups": n_late}))


if __name__ == "__main__":
    main()
