#!/usr/bin/env python3
"""Where does the verification of a short pair list spend its wall time?  From a rocprofv3 --kernel-trace CSV of a bench.py
run (last step): the span of the verification, the summed kernel time per stream-agnostic timeline, and the GAPS between
consecutive kernels (end of the latest-ending kernel so far -> start of the next) bucketed by length -- a host round trip
(hipMemcpyAsync + hipStreamSynchronize + the next launch) shows up as a gap of tens of microseconds, back-to-back launches
of one stream as a few.  Usage: python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [s for s, e, n in ev if "k1_best_rows<false>" in n or "k1_best_rows<(bool)0>" in n]
t0 = starts[-1]
ev = [x for x in ev if x[0] >= t0]
first_verify = min(s for s, e, n in ev if "k_verify_prep" in n)
# the verification ends with its last k_final_finish / k_compact_inliers; what follows (result fetches, torch kernels of the
# statistics, the next step) is not part of it
last_final = max(e for s, e, n in ev if s >= first_verify and ("k_final_finish" in n or "k_compact_inliers" in n))
ver = [x for x in ev if x[0] >= first_verify and x[0] <= last_final]
t1 = max(e for s, e, n in ver)
busy_end = first_verify
gaps = []
busy = 0
for s, e, n in ver:
    if s > busy_end:
        gaps.append((s - busy_end, n))
        busy_end_prev = busy_end
    if e > busy_end:
        busy += e - max(s, busy_end)
        busy_end = e
print("verification span %.2f ms over %d kernels: some kernel running %.2f ms, nothing running %.2f ms (%d gaps)" %
      ((t1 - first_verify) / 1e6, len(ver), busy / 1e6, sum(g for g, _ in gaps) / 1e6, len(gaps)))
for lo, hi in ((0, 5), (5, 15), (15, 40), (40, 100), (100, 1000), (1000, 10 ** 9)):
    sel = [g for g, _ in gaps if lo * 1000 <= g < hi * 1000]
    print("  gaps %4d-%-6s us: %4d  total %7.3f ms" % (lo, hi if hi < 10 ** 9 else "inf", len(sel), sum(sel) / 1e6))
after = {}
for g, n in gaps:
    k = n.split("(")[0][:50]
    a = after.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += g
print("gaps by the kernel that FOLLOWS them (top 12):")
for k, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-52s %4d gaps  %7.3f ms" % (k, c, t / 1e6))
