#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace CSV for the LAST step in it (from the dispatch after the second-to-last
k_verify_final to the last k_verify_final): per kernel calls / busy time, and the idle time between consecutive
dispatches (host round trips, launch latency)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
finals = [i for i, e in enumerate(ev) if e[2].startswith("k_verify_final")]
last = finals[-1]
first = finals[-2] + 1 if len(finals) > 1 else 0
seg = ev[first:last + 1]
t0, t1 = seg[0][0], seg[-1][1]
busy = defaultdict(lambda: [0, 0])
idle = 0
prev_end = seg[0][0]
for s, e, n in seg:
    name = n.split("(")[0][:44]
    busy[name][0] += 1
    busy[name][1] += e - s
    if s > prev_end:
        idle += s - prev_end
    prev_end = max(prev_end, e)
print("span %.2f ms, dispatches %d, idle between dispatches %.2f ms" % ((t1 - t0) / 1e6, len(seg), idle / 1e6))
for name, (c, d) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-46s %5d %9.3f ms" % (name, c, d / 1e6))
