#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of a bench.py run: for the LAST step (from the last k1_best_rows<false>
launch on), the span, the time during which NO kernel was running on any stream (host round trips, launch gaps), the time
during which only `small` kernels (grid below 256 workgroups = less than one per CU) were running, and the per-kernel
totals.  Usage: python tools/trace_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 64)) or 64)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1))
    ev.append((s, e, name, grid // max(wg, 1)))
ev.sort()
starts = [s for s, e, n, g in ev if "k1_best_rows<false>" in n or "k1_best_rows<(bool)0>" in n]
t0 = starts[-1]
ev = [x for x in ev if x[0] >= t0]
t1 = max(e for s, e, n, g in ev)
first_verify = min(s for s, e, n, g in ev if "k_verify_prep" in n)
print("last step: %.2f ms total; matching part %.2f ms; verification part %.2f ms" % ((t1 - t0) / 1e6, (first_verify - t0) / 1e6, (t1 - first_verify) / 1e6))
# sweep over the verification part
pts = []
for s, e, n, g in ev:
    if e <= first_verify:
        continue
    big = g >= 256
    pts.append((max(s, first_verify), 1, big))
    pts.append((e, -1, big))
pts.sort()
idle = small_only = 0
n_all = n_big = 0
last = first_verify
for t, d, big in pts:
    if t > last:
        if n_all == 0:
            idle += t - last
        elif n_big == 0:
            small_only += t - last
        last = t
    n_all += d
    if big:
        n_big += d
print("verification: no kernel running %.2f ms; only kernels with < 256 workgroups running %.2f ms; at least one larger kernel %.2f ms" %
      (idle / 1e6, small_only / 1e6, (t1 - first_verify - idle - small_only) / 1e6))
tot = defaultdict(lambda: [0, 0, 0])
for s, e, n, g in ev:
    k = n.split("(")[0][:60]
    tot[k][0] += 1
    tot[k][1] += e - s
    tot[k][2] += (e - s) if g < 256 else 0
for k, (c, t, ts) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-62s calls %5d  %8.2f ms  (of which with < 256 workgroups %7.2f ms)" % (k, c, t / 1e6, ts / 1e6))
