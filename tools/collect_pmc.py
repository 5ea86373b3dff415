#!/usr/bin/env python3
"""Collects PMC counters of the matching kernels (k1_best_rows<false> = pass 1, <true> = gathered pass 2) with
rocprofv3, in separate passes as MI355X_MICROARCH.md (section HBM / rocprofv3 PMC slots) prescribes: FETCH_SIZE
costs 3 TCC slots and WRITE_SIZE 2, so they cannot share a pass; the SQ utilisation counters get passes of their
own.  Every pass is `rocprofv3 --pmc <counters> --kernel-trace` only (no other trace domain).

On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced (16 B/lane) streaming read -- K1's
B-tile and A-fragment loads are 16 B/lane -- so the read side is doubled; WRITE_SIZE is uncalibrated and reported
as is.  Units: KiB per dispatch.

Run on the GPU box:  python tools/collect_pmc.py [--util] [bench args]   ->  gpurun_out/pmc/k1_pmc.json"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "gpurun_out", "pmc")
os.makedirs(out_dir, exist_ok=True)
argv = sys.argv[1:]
util = "--util" in argv
util1 = "--util1" in argv  # only the MFMA / VALU instruction and busy counters
tag_out = None
if "--out" in argv:
    tag_out = argv[argv.index("--out") + 1]
    del argv[argv.index("--out"):argv.index("--out") + 2]
argv = [a for a in argv if a not in ("--util", "--util1")]
bench_args = argv or ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0"]
res = {"bench_args": bench_args}
env = dict(os.environ, TMPDIR="/tmp")
passes = [("FETCH_SIZE",), ("WRITE_SIZE",)]
if util1:
    passes += [("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_I8", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU",
                "SQ_INSTS_LDS", "SQ_WAVE_CYCLES")]
if util:
    passes += [("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_I8"),
               ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"),
               ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS")]


def kname(row):
    n = row.get("Kernel_Name", "")
    if "k1_best_rows" not in n:
        return None
    return "pass2" if "<true>" in n else "pass1"


for counters in passes:
    tag = counters[0]
    d = os.path.join(out_dir, tag)
    cmd = ["rocprofv3", "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                                     sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    with open(os.path.join(out_dir, tag + ".log"), "w") as log:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=False)
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):  # durations of the same dispatches
        for row in csv.DictReader(open(f)):
            k = kname(row)
            if k is not None:
                res.setdefault("duration_ms_" + tag, {}).setdefault(k, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = kname(row)
            if k is None or row.get("Counter_Name") not in counters:
                continue
            acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (k, c), vals in sorted(acc.items()):
        res.setdefault(c, {})[k] = {"dispatches": len(vals), "mean_per_dispatch": sum(vals) / len(vals)}


def _arg(name, default):
    return int(bench_args[bench_args.index(name) + 1]) if name in bench_args else default


res["images"], res["feats"] = _arg("--images", 500), _arg("--feats", 4096)
res["pairs"] = res["images"] * (res["images"] - 1) // 2 if "--pairs" not in bench_args else None
res["variant"] = "dot4 (DSM_K1_DOT4)" if os.environ.get("DSM_K1_DOT4") else "mfma"
f, w = res.get("FETCH_SIZE", {}), res.get("WRITE_SIZE", {})
if f and w:
    # per step (one launch of each pass): FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as reported; KiB -> bytes
    tot = 0.0
    for k in ("pass1", "pass2"):
        if k in f:
            tot += 2.0 * f[k]["mean_per_dispatch"]
        if k in w:
            tot += w[k]["mean_per_dispatch"]
    res["k1_traffic_bytes_per_launch"] = tot * 1024.0
    res["algorithmic_bytes_per_launch"] = None if res["pairs"] is None else res["pairs"] * 2.0 * res["feats"] * 128.0
if util and "SQ_VALU_MFMA_BUSY_CYCLES" in res and "SQ_BUSY_CU_CYCLES" in res:
    for k in ("pass1", "pass2"):
        try:
            res.setdefault("mfma_busy_ratio", {})[k] = (res["SQ_VALU_MFMA_BUSY_CYCLES"][k]["mean_per_dispatch"] /
                                                        (4.0 * res["SQ_BUSY_CU_CYCLES"][k]["mean_per_dispatch"]))
        except Exception:
            pass
print(json.dumps(res))
json.dump(res, open(os.path.join(out_dir, "k1_pmc.json"), "w"), indent=1)
if tag_out:
    json.dump(res, open(tag_out, "w"), indent=1)
