#!/usr/bin/env python3
"""Coarse timeline of the last step in a rocprofv3 --kernel-trace CSV: per time bucket and per queue, the kernel that
was busy longest, plus the fraction of the bucket in which any kernel ran on that queue."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
bucket_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:22], r["Queue_Id"]) for r in rows))
finals = [i for i, e in enumerate(ev) if e[2].startswith("k_verify_final")]
t_end = ev[finals[-1]][1]
# the step starts at the last k1_best_rows<false> before it
starts = [e[0] for e in ev if e[2].startswith("k1_best_rows<false>") and e[0] < t_end]
t0 = starts[-1]
seg = [e for e in ev if e[0] >= t0 and e[1] <= t_end]
queues = sorted({e[3] for e in seg})
B = int(bucket_ms * 1e6)
nb = (t_end - t0) // B + 1
print("step %.1f ms, queues %s" % ((t_end - t0) / 1e6, queues))
for b in range(nb):
    lo, hi = t0 + b * B, t0 + (b + 1) * B
    line = "%6.0f ms " % (b * bucket_ms)
    for q in queues:
        busy = defaultdict(int)
        for s, e, n, qq in seg:
            if qq != q or e <= lo or s >= hi:
                continue
            busy[n] += min(e, hi) - max(s, lo)
        tot = sum(busy.values())
        top = max(busy.items(), key=lambda kv: kv[1])[0] if busy else "-"
        line += "| q%s %3.0f%% %-22s " % (q, 100.0 * tot / B, top)
    print(line)
