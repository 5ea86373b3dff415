#!/usr/bin/env python3
"""Collects HBM traffic of the dominant kernel (k1_best_rows) with rocprofv3 PMC counters, in separate
passes as MI355X_MICROARCH.md (section HBM / rocprofv3 PMC slots) prescribes: FETCH_SIZE costs 3 TCC slots
and WRITE_SIZE 2, so they cannot share a pass.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced (16 B/lane) streaming read -- K1's B-tile and A-fragment loads are 16 B/lane -- so the read
side is doubled; WRITE_SIZE is uncalibrated and reported as is.  Units: KiB per dispatch.
Run on the GPU box:  python tools/collect_pmc.py [bench args]   ->  gpurun_out/pmc/k1_pmc.json"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "gpurun_out", "pmc")
os.makedirs(out_dir, exist_ok=True)
bench_args = sys.argv[1:] or ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0"]
res = {"bench_args": bench_args}
env = dict(os.environ, TMPDIR="/tmp")
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out_dir, counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    with open(os.path.join(out_dir, counter + ".log"), "w") as log:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=False)
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k1_best_rows" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    res[counter] = {"dispatches": len(vals), "sum_per_dispatch_KiB": (sum(vals) / max(len(set(range(len(vals)))), 1)) if vals else None,
                    "values": vals[:8]}
def _arg(name, default):
    return int(bench_args[bench_args.index(name) + 1]) if name in bench_args else default


res["images"], res["feats"] = _arg("--images", 500), _arg("--feats", 4096)
f, w = res["FETCH_SIZE"]["values"], res["WRITE_SIZE"]["values"]
if f and w:
    # per launch: FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; KiB -> bytes
    res["k1_traffic_bytes_per_launch"] = (2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0
print(json.dumps(res))
json.dump(res, open(os.path.join(out_dir, "k1_pmc.json"), "w"), indent=1)
