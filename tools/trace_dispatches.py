#!/usr/bin/env python3
"""Per-dispatch view of the LAST step in a rocprofv3 --kernel-trace CSV: every launch whose name contains one of the
given substrings, in start order per queue, with its grid (workgroups), duration and the gap to the launch before it on
the same queue.  Shows what ONE launch of a chain costs (the replay / local-optimisation iterations of a round), which the
per-kernel totals of trace_summary.py hide.
Usage: python tools/trace_dispatches.py <kernel_trace.csv> [substring ...]   (default: k_replay_lo)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2:] or ["k_replay_lo"]
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 64)) or 64)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    ev.append((s, e, name, grid // max(wg, 1), r.get("Queue_Id", "0")))
ev.sort()
starts = [s for s, e, n, g, q in ev if "k1_best_rows<false>" in n or "k1_best_rows<(bool)0>" in n]
t0 = starts[-1]
ev = [x for x in ev if x[0] >= t0]
last_end = defaultdict(lambda: t0)
tot = defaultdict(lambda: [0, 0.0])
for s, e, n, g, q in ev:
    gap = (s - last_end[q]) / 1e3
    last_end[q] = e
    if any(x in n for x in subs):
        print("q%-3s t=%9.3f ms  %-34s wgs %6d  dur %9.1f us  gap-before %8.1f us" % (q, (s - t0) / 1e6, n[:34], g, (e - s) / 1e3, gap))
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e6
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("TOTAL %-40s calls %4d  %8.2f ms" % (n[:40], c, t))
