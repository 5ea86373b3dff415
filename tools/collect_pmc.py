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
verify_mode = "--verify" in argv  # every kernel of the step by name (the verification kernels), executed-instruction counters
util = "--util" in argv
util1 = "--util1" in argv  # only the MFMA / VALU instruction and busy counters
tag_out = None
if "--out" in argv:
    tag_out = argv[argv.index("--out") + 1]
    del argv[argv.index("--out"):argv.index("--out") + 2]
argv = [a for a in argv if a not in ("--util", "--util1", "--verify")]
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


if verify_mode:
    # SQ counters only (<= 8 per pass).  FP64 instruction classes: the verification is built with -ffp-contract=off, so its
    # arithmetic is ADD_F64 / MUL_F64 (1 flop per lane), FMA_F64 appears only inside the division / square-root expansions,
    # TRANS_F64 is v_rcp / v_rsq / v_sqrt_f64 (quarter rate).
    passes = [("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_THREAD_CYCLES_VALU"),
              ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU",
               "SQ_INSTS_VALU_INT32"),
              ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM",
               "SQ_ACTIVE_INST_ANY")]


def kname(row):
    n = row.get("Kernel_Name", "")
    if verify_mode:
        n = n.replace("void ", "").split("(")[0].strip()
        return n if n and not n.startswith("k1_") and not n.startswith("__amd") else None
    if "k1_best_rows" not in n:
        return None
    return "pass2" if "<true>" in n else "pass1"


def summarize_verify(res):
    """Per kernel: time, VALU issue utilisation, the FP64 instruction mix and the executed FP64 rate against the vector peak
    (256 CUs x 4 SIMDs x 16 lanes x clock: one wave64 FP64 instruction = 4 cycles of a SIMD; an FMA counts 2 flops)."""
    def tot(counter, k):  # summed over the dispatches of the step
        v = res.get(counter, {}).get(k)
        return v["mean_per_dispatch"] * v["dispatches"] if v else 0.0
    cus, clk = 256, 2.4e9
    issue_peak = cus * 4 * clk / 4.0        # wave64 VALU instructions per second, whole chip
    fp64_peak = cus * 4 * 16 * 2 * clk      # 78.6 TFLOP/s
    names = sorted(res.get("SQ_INSTS_VALU", {}).keys())
    dur = {}
    for k, v in res.get("duration_ms_SQ_INSTS_VALU", {}).items():
        dur[k] = sum(v)
    rows = {}
    for k in names:
        ms = dur.get(k, 0.0)
        if ms <= 0:
            continue
        valu = tot("SQ_INSTS_VALU", k)
        f64 = {c: tot("SQ_INSTS_VALU_" + c + "_F64", k) for c in ("ADD", "MUL", "FMA", "TRANS")}
        f64_insts = sum(f64.values())
        flops = 64.0 * (f64["ADD"] + f64["MUL"] + 2.0 * f64["FMA"] + f64["TRANS"])
        busy_cu = tot("SQ_BUSY_CU_CYCLES", k)
        wave_cyc = tot("SQ_WAVE_CYCLES", k)
        rows[k] = {
            "ms_per_step": ms, "dispatches": res["SQ_INSTS_VALU"][k]["dispatches"],
            "valu_insts": valu, "fp64_insts": f64_insts, "fp64_mix": f64,
            "fp64_share_of_valu": f64_insts / valu if valu else None,
            "valu_issue_util": valu / (ms * 1e-3) / issue_peak,                  # of the nameplate clock
            "executed_fp64_tflops": flops / (ms * 1e-3) / 1e12,
            "executed_frac": flops / (ms * 1e-3) / fp64_peak,
            "lane_util": (tot("SQ_THREAD_CYCLES_VALU", k) / (64.0 * tot("SQ_ACTIVE_INST_VALU", k))) if tot("SQ_ACTIVE_INST_VALU", k) else None,
            "clock_ghz_while_busy": busy_cu / cus / (ms * 1e-3) / 1e9 if busy_cu else None,
            # SQ_WAVE_CYCLES counts QUAD-cycles (MI355X_MICROARCH.md, "SQ PMC units"), SQ_BUSY_CU_CYCLES cycles: resident waves per CU =
            # 4 wave_cyc / busy_cu, per SIMD a quarter of that.  (Files collected before the last commit of round 6 divided by four once
            # too often: their figure is a quarter of the truth -- 0.25 there is ONE wave per SIMD, 0.5 two.)
            "waves_per_simd_avg": wave_cyc / busy_cu if busy_cu else None,
            "wait_inst_any_share": tot("SQ_WAIT_INST_ANY", k) / wave_cyc if wave_cyc else None,
            "lds_insts": tot("SQ_INSTS_LDS", k), "lds_bank_conflict_cycles": tot("SQ_LDS_BANK_CONFLICT", k),
            "salu_insts": tot("SQ_INSTS_SALU", k), "vmem_insts": tot("SQ_INSTS_VMEM", k),
        }
    total_ms = sum(r["ms_per_step"] for r in rows.values())
    total_flops = sum(r["executed_fp64_tflops"] * r["ms_per_step"] for r in rows.values())
    sc = [r for k, r in rows.items() if k.startswith(("k_prescore", "k_score", "k_models_score"))]
    sc_ms = sum(r["ms_per_step"] for r in sc)
    res["summary"] = {"kernels": dict(sorted(rows.items(), key=lambda kv: -kv[1]["ms_per_step"])),
                      "scoring_kernels_ms_per_step": sc_ms,
                      "scoring_kernels_executed_frac": (sum(r["executed_frac"] * r["ms_per_step"] for r in sc) / sc_ms) if sc_ms else None,
                      # (the bound steps classify in PACKED f32 since rounds 5 / 6: the FP64 counters do not see that work, the issue rate does)
                      "scoring_kernels_valu_issue": (sum(r["valu_issue_util"] * r["ms_per_step"] for r in sc) / sc_ms) if sc_ms else None,
                      "all_kernels_ms_per_step": total_ms,
                      "executed_fp64_tflops_over_all": total_flops / total_ms if total_ms else None,
                      "executed_frac_over_all": total_flops / total_ms * 1e12 / fp64_peak if total_ms else None,
                      "peaks": {"fp64_vector_tflops": fp64_peak / 1e12, "valu_wave_insts_per_s": issue_peak},
                      "note": "durations and counters of ONE step under counter collection (kernels serialised, one lane: "
                              "DSM_VERIFY_LANES=1); SQ counters summed over all CUs"}


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
# what the collection depends on: bench.py refuses a committed profile whose hash is not that of the sources it runs with
sys.path.insert(0, ROOT)
import bench  # noqa: E402
res["source_hash"] = {k: bench.source_hash(ROOT, k) for k in bench.PROFILE_SOURCES}
res["commit"] = os.environ.get("DSM_COMMIT")  # the GPU box has no .git: the session script passes `git rev-parse --short HEAD` of the snapshot
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
if verify_mode:
    summarize_verify(res)
print(json.dumps(res if not verify_mode else res.get("summary")))
json.dump(res, open(os.path.join(out_dir, "k1_pmc.json" if not verify_mode else "verify_pmc.json"), "w"), indent=1)
if tag_out:
    json.dump(res, open(tag_out, "w"), indent=1)
