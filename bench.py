#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: verified image-pairs/s (+ RANSAC hypotheses/s) at 4 096 feats/image.

One "step" = one pass of the hot path (brute-force matching + two-view verification + assembly of the match
graph) over the whole pair list of the workload, inputs already resident in HBM.  Default workload = BASELINE.json
configs[1]: 500 images x 4 096 features, exhaustive => 124 750 pairs, calibrated (E + F + H + relative pose).
With N GPUs the pair list is block-partitioned over the ranks (strong scaling: the workload is fixed), every rank
runs the same kernels on its share, and the per-pair results are all-gathered with RCCL so that every rank holds
the full match graph (dagsfm_amd/sharding.py).

  python bench.py --gpus N --steps K --warmup W          # N > 1: re-launches itself under torch.distributed.run
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Other workloads of BASELINE.json (parity-test / profile cases, not the default line):
  --images 50 --feats 1024 --uncalibrated                          configs[0] shape on the GPU
  --images 2000 --no-verify                                         configs[2]
  --images 10000 --pairs knn:200 --shard-of 8                       configs[3], one GPU's shard of the 8
  --images 10000 --feats 8192 --pairs knn:200 --fixed-trials 4096 --shard-of 8 --max-pairs N    configs[4], shard

Rank 0 prints ONE JSON line (README / DESIGN.md "Measurement").
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Sanity bounds from /opt/skills/guides/MI355X_MICROARCH.md (the peaks themselves are derived from hipDeviceProp).
GUIDE_INT8_DENSE = 5.0e15   # ops/s: "I8 ~2x bf16 rate", bf16 dense ~2.5 PF
GUIDE_FP64_VECTOR = 78.6e12
GUIDE_HBM = 8.0e12


LINE_LIMIT = 4096          # the driver reads the last 8 KB of stdout; the line must fit with room to spare
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


def _round_floats(x, sig=6):
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _round_floats(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round_floats(v, sig) for v in x]
    return x


# keys that only the --dump-line file carries
LONG_FORM_ONLY = ("peak_source", "sample_note", "top5_ms_execfrac_laneutil", "ranks_seen_by_process_group", "stale_files", "side_pose_max_rel",
                  "source_hash", "executed_fp64_tflops_over_all", "traffic_file", "avg_launch_ms")


def _drop_notes(x):
    """The prose (what each figure means) lives in README.md "Measurement" and in the --dump-line file, not on stdout."""
    if isinstance(x, dict):
        return {k: _drop_notes(v) for k, v in x.items()
                if not (k == "note" or k.endswith("_note") or k in LONG_FORM_ONLY)}
    return x


def format_line(out, dump_path="", limit=LINE_LIMIT):
    """The ONE stdout line: `out` without its prose and with floats at 6 significant digits.  The long form goes to
    `dump_path`.  Fails loudly instead of printing a line the driver's 8 KB tail would cut."""
    if dump_path:
        with open(dump_path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    for k in REQUIRED_KEYS:
        if k not in out:
            raise SystemExit("bench.py: result lacks the contract key %r" % k)
    if "workload" not in out["config"]:
        raise SystemExit("bench.py: config.workload missing")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        if k not in out["roofline"]:
            raise SystemExit("bench.py: roofline.%s missing" % k)
    line = json.dumps(_round_floats(_drop_notes(out)), separators=(",", ":"))
    if len(line) >= limit:
        raise SystemExit("bench.py: the result line is %d bytes (limit %d): move detail to --dump-line" % (len(line), limit))
    return line


PROFILE_SOURCES = {  # the sources a counter collection depends on (tools/collect_pmc.py stamps their hash into the file)
    "k1": ("match_kernels.hip",),
    "verify": ("verify_kernels.hip", "verify_linalg.h", "verify_estimators.h", "verify_camera.h", "verify_fivept_coop.h", "fivept_terms.tbl"),
}


def source_hash(root, which):
    import hashlib
    h = hashlib.sha256()
    for name in PROFILE_SOURCES[which]:
        with open(os.path.join(root, "dagsfm_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def profile_figures(root, images, feats, n_pairs, verify_too):
    """Figures READ FROM COMMITTED FILES under profiles/ (counter collections of an earlier run of the same workload,
    tools/collect_pmc.py) -- never measured in this run, so they sit under one `from_profiles` key with their file names, the
    commit they were collected at and the hash of the kernel sources they were collected from.  A file whose hash is not that of
    the sources in this tree is REFUSED (a stale profile must not speak for a changed kernel): it is named
    under `stale` and contributes nothing."""
    import glob
    fp = {}
    stale = []

    def fresh(pmc, f, which):
        if pmc.get("source_hash", {}).get(which) == source_hash(root, which):
            return True
        stale.append("profiles/" + os.path.basename(f))
        return False
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r0[3-9]_k1_pmc*.json"))):
        try:
            pmc = json.load(open(f))
        except Exception:
            continue
        if pmc.get("images") == images and pmc.get("feats") == feats and pmc.get("pairs") == n_pairs and fresh(pmc, f, "k1"):
            mi = pmc.get("SQ_INSTS_VALU_MFMA_I8", {})
            insts = None
            if "pass1" in mi:
                insts = mi["pass1"]["mean_per_dispatch"] + mi.get("pass2", {}).get("mean_per_dispatch", 0.0)
            fp["k1"] = {"file": "profiles/" + os.path.basename(f), "commit": pmc.get("commit"), "source_hash": pmc["source_hash"]["k1"],
                        "hbm_bytes_per_launch": pmc.get("k1_traffic_bytes_per_launch"), "mfma_i8_insts_per_launch": insts}
    if verify_too:
        for f in sorted(glob.glob(os.path.join(root, "profiles", "r0[4-9]_verify_pmc.json"))):
            try:
                vp = json.load(open(f))
            except Exception:
                continue
            if vp.get("images") == images and vp.get("feats") == feats and vp.get("pairs") == n_pairs and fresh(vp, f, "verify"):
                sm = vp.get("summary", {})
                ks = sorted(sm.get("kernels", {}).items(), key=lambda kv: -kv[1].get("ms_per_step", 0.0))[:5]
                fp["verify"] = {"file": "profiles/" + os.path.basename(f), "commit": vp.get("commit"), "source_hash": vp["source_hash"]["verify"],
                                "all_kernels_ms_per_step": sm.get("all_kernels_ms_per_step"),
                                "executed_fp64_tflops_over_all": sm.get("executed_fp64_tflops_over_all"),
                                "executed_frac_over_all": sm.get("executed_frac_over_all"),
                                "scoring_kernels_frac": sm.get("scoring_kernels_executed_frac"), "scoring_kernels_valu_issue": sm.get("scoring_kernels_valu_issue"),
                                "top5_ms_execfrac_laneutil": {k: [v.get("ms_per_step"), v.get("executed_frac"), v.get("lane_util")] for k, v in ks}}
    if stale:
        fp["stale"] = len(set(stale))  # committed collections of OTHER sources than this tree's: refused
        fp["stale_files"] = sorted(set(stale))
    return fp


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pairs", default="exhaustive", help="exhaustive | knn:K (K pseudo-neighbours per image from a seeded "
                                                          "kNN over the camera centres, id1<id2 dedupe; SURVEY 8d config 4)")
    ap.add_argument("--cut", default="contiguous", choices=["contiguous", "interleaved"],
                    help="how the pair list is cut between the ranks (and by --shard-of): one contiguous cost-balanced stretch per rank, or "
                         "blocks of 256 pairs dealt out round-robin / heaviest-first (sharding.interleaved_parts): every rank a uniform sample "
                         "of the list, at the price of putting the gathered results back into list order (1.5 - 1.8 ms at config 2).  On the "
                         "synthetic exhaustive list the two have the same slowest shard (profiles/r06_shard_sweep_config2.txt), so the default "
                         "stays the cut without a reorder")
    ap.add_argument("--shard-of", type=int, default=1, help="run only one 1/S of the pair list: one GPU's shard of an S-GPU config")
    ap.add_argument("--shard-index", type=int, default=0, help="which of the --shard-of shards (0-based): tools/shard_sweep.py runs them all")
    ap.add_argument("--planar", action="store_true", help="a planar scene (H is the model: its local optimisations have hundreds of inliers); not the driver's workload")
    ap.add_argument("--outlier-frac", type=float, default=0.2,
                    help="share of an image's features that are not observations of the scene: a putative match is geometrically "
                         "right with (1 - f)^2; 0.2 = the headline's 0.64 inlier ratio, 0.5 = a 0.25 ratio (RANSAC needs ~13x the trials)")
    ap.add_argument("--dump-graph", default="", help="rank 0 saves the assembled match graph of the last step to this .npz (tests)")
    ap.add_argument("--max-pairs", type=int, default=0, help="truncate the (sharded) pair list (bounded runs of the big configs)")
    ap.add_argument("--fixed-trials", type=int, default=0,
                    help="T > 0: min_num_trials = max_num_trials = T, confidence 0.999999, min_inlier_ratio 0.01 -- exactly T "
                         "trials per family and pair (SURVEY 8d config 5)")
    ap.add_argument("--no-verify", action="store_true", help="matching only (BASELINE configs[2])")
    ap.add_argument("--uncalibrated", action="store_true",
                    help="cameras without focal prior: F + H path (EstimateUncalibrated) instead of E + F + H + pose")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU-baseline samples (0 = skip)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="dsm contexts per GPU: the rank's share of the pair list is cut into that many contiguous parts, each "
                         "matched + verified by its own context / stream / host thread (the latency-bound tails of one part's "
                         "RANSAC rounds overlap with the bulk kernels of another)")
    ap.add_argument("--force-collectives", nargs="?", const="auto", default="", choices=["", "auto", "padded", "broadcast"],
                    help="one rank only: form a 1-rank RCCL process group and send the results through the same all-gather / "
                         "broadcast calls the N-rank exchange makes (the part of the RCCL path a one-GPU box can run); the "
                         "line then carries the measured exchange time")
    ap.add_argument("--no-second-regime", action="store_true",
                    help="skip the low-inlier-ratio side measurement (extra.low_inlier_regime) after the timed region")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the 2 000-image matching-only side measurement (extra.config3_match_only: BASELINE configs[2], the shape north_star's "
                         ">= 10x the host CPU on 2 000-image exhaustive matching target is stated on) after the timed region")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the two further side measurements after the timed region: the same pair list with cameras without a focal prior "
                         "(extra.uncalibrated: F + H) and BASELINE configs[0] on the GPU (extra.config1: 50 x 1 024, every pair checked)")
    ap.add_argument("--memory-budget-gib", type=float, default=0.0,
                    help="dsm_ctx_set_memory_budget: GiB of transient chunk scratch the matcher and the verifier may hold (0: the defaults)")
    ap.add_argument("--no-match-lock", action="store_true", help="experiment: with --contexts > 1, let the contexts' matching calls overlap")
    ap.add_argument("--ctx-after-pg", action="store_true", help="experiment: create the dsm contexts after the process group (the order of rounds 1 - 4)")
    ap.add_argument("--dump-line", default="", help="rank 0 also writes the long form of the result (with the prose notes) to this file")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="debug: all ranks on device 0 over gloo (exercises the multi-rank path on a 1-GPU box)")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_multi_rank(args):
    """`python bench.py --gpus N` without a launcher: become the launcher (one process per GPU, RCCL)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(orc, label, build, scene_images, pairs, budget_s, verify, cams, opts, user_seed, cores, keep=None, every_pair=False):
    """Times a CPU oracle build (the reference algorithm restated, oracle/) on a bounded sample of the same
    workload: worker threads pull evenly spaced pairs of the list until `budget_s` seconds have passed, one thread
    per host core like the reference's matcher/verifier thread pools (/root/reference/src/feature/matching.cc:640-674).
    `keep` (a dict): the oracle's result of every sampled pair -- pair index -> (matches, TwoViewGeometry, inlier matches) --
    for parity_sample(); `every_pair`: the sample is the whole list in order (small workloads)."""
    import threading
    from dagsfm_amd import capi
    kps = [im[1].astype(np.float64) if (verify and im is not None) else None for im in scene_images]
    order = np.arange(len(pairs), dtype=np.int64) if every_pair else np.linspace(0, len(pairs) - 1, min(len(pairs), 65536)).astype(np.int64)
    lock = threading.Lock()
    state = {"next": 0, "pairs": 0, "models": 0}
    deadline = time.perf_counter() + budget_s

    def worker():
        while time.perf_counter() < deadline:
            with lock:
                k = state["next"]
                state["next"] += 1
            if k >= len(order):
                return
            i, j = int(pairs[order[k]][0]), int(pairs[order[k]][1])
            m = orc.match_sift_features_cpu(scene_images[i][0], scene_images[j][0])
            nm, tv, inl = 0, None, None
            if verify:
                tv, inl = orc.estimate_two_view_geometry(cams[i], kps[i], cams[j], kps[j], m, opts, capi.pair_seed(i, j, user_seed))
                nm = sum(tv.num_models)
            with lock:
                state["pairs"] += 1
                state["models"] += nm
                if keep is not None:
                    keep[int(order[k])] = (m, tv, inl)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker) for _ in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": state["pairs"] / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "build": build,
            "hypotheses_per_s": state["models"] / dt,
            "sample": "%d of %d pairs, %s (%s), %.1f s on %d threads"
                      % (state["pairs"], len(pairs), "all" if every_pair else "evenly spaced", "match only" if not verify else "match + verify", dt, cores),
            "sample_note": "host has %d usable cores; oracle/ = the reference CPU path restated (MatchSiftFeaturesCPU + "
                           "TwoViewGeometry::Estimate), %s" % (host_cores(), label)}


class GraphView:
    """Per-pair slices of an assembled MatchGraph (torch tensors on the device, sharding.MatchGraph) or of a context's own results
    (numpy arrays of the C-ABI getters) on the host -- only the sampled pairs are copied."""

    def __init__(self, match_counts=None, matches=None, tvg=None, inlier_counts=None, inlier_matches=None, graph=None):
        import torch
        if graph is not None:
            match_counts, matches, tvg, inlier_counts, inlier_matches = graph.match_counts, graph.matches, graph.tvg, graph.inlier_counts, graph.inlier_matches

        def offs(c):
            c = torch.as_tensor(c).reshape(-1).to(torch.int64)
            return torch.cat([torch.zeros(1, dtype=torch.int64, device=c.device), torch.cumsum(c, 0)]).cpu().numpy()
        self.moff = offs(match_counts)
        self.matches = matches
        self.tvg = tvg
        self.ioff = offs(inlier_counts) if inlier_counts is not None else None
        self.inl = inlier_matches

    @staticmethod
    def _rows(a, lo, hi):
        r = a[int(lo):int(hi)]
        return (r.cpu().numpy() if hasattr(r, "cpu") else np.asarray(r)).astype(np.uint32).reshape(-1, 2)

    def pair_matches(self, k):
        return self._rows(self.matches, self.moff[k], self.moff[k + 1])

    def pair_inliers(self, k):
        return self._rows(self.inl, self.ioff[k], self.ioff[k + 1])

    def pair_tvg(self, k):
        from dagsfm_amd import capi
        r = self.tvg[k]
        if isinstance(r, capi.TwoViewGeometry):
            return r
        b = (r.cpu().numpy() if hasattr(r, "cpu") else np.asarray(r)).tobytes()
        return capi.TwoViewGeometry.from_buffer_copy(b)


def parity_sample(keep, view, verify, min_num_inliers, stage_filter=True):
    """Holds the device's results -- the graph the timed region just assembled, or a side measurement's -- against the oracle's
    results of the SAME pairs (what cpu_baseline computed while it was being timed): match lists index by index
    (/root/reference/src/feature/sift.cc:164-198), and per pair the TwoViewGeometry decision for decision -- config, inlier count,
    trial and model counts, E / F / H bit for bit, the inlier matches; qvec / tvec / tri_angle within 1e-6 relative
    (/root/reference/src/estimators/two_view_geometry.cc:232-425), the bar of tests/test_verify_gpu.py::tvg_equal.  A pair the
    stage's post-filter empties (fewer than min_num_inliers inliers, matching.cc:824-831) must be empty on the device."""
    mm = gm = 0
    pose = 0.0
    first_bad = None
    for k in sorted(keep):
        m, tv, inl = keep[k]
        g = view.pair_matches(k)
        ok_m = g.shape == np.asarray(m).reshape(-1, 2).shape and bool((g == np.asarray(m).reshape(-1, 2)).all())
        if not ok_m:
            mm += 1
            first_bad = first_bad if first_bad is not None else int(k)
        if not verify or tv is None:
            continue
        d = view.pair_tvg(k)
        di = view.pair_inliers(k)
        if stage_filter and tv.num_inliers < min_num_inliers:
            ok = d.config == 0 and d.num_inliers == 0 and len(di) == 0
        else:
            ok = (d.config == tv.config and d.num_inliers == tv.num_inliers and d.num_matches == tv.num_matches
                  and list(d.num_trials) == list(tv.num_trials) and list(d.num_models) == list(tv.num_models))
            for name in ("E", "F", "H"):
                ok = ok and np.array_equal(np.array(getattr(d, name)), np.array(getattr(tv, name)), equal_nan=True)
            ok = ok and di.shape == np.asarray(inl).reshape(-1, 2).shape and bool((di == np.asarray(inl).reshape(-1, 2)).all())
            for name in ("qvec", "tvec"):
                a, b = np.array(getattr(d, name)), np.array(getattr(tv, name))
                ok = ok and bool(np.allclose(a, b, rtol=1e-6, atol=1e-12))
                if np.isfinite(a).all() and np.isfinite(b).all() and np.abs(b).max() > 0:
                    pose = max(pose, float(np.abs(a - b).max() / np.abs(b).max()))
            ok = ok and abs(d.tri_angle - tv.tri_angle) <= 1e-6 * max(abs(tv.tri_angle), 1e-9)
            if tv.tri_angle != 0 and np.isfinite(tv.tri_angle):
                pose = max(pose, abs(d.tri_angle - tv.tri_angle) / abs(tv.tri_angle))
        if not ok:
            gm += 1
            first_bad = first_bad if first_bad is not None else int(k)
    out = {"pairs": len(keep), "match_mismatches": mm}
    if verify:
        out["geometry_mismatches"] = gm
        out["pose_max_rel"] = pose
    if first_bad is not None:
        out["first_bad_pair"] = first_bad
    return out


def parity_failed(ps):
    return bool(ps) and (ps.get("match_mismatches", 0) != 0 or ps.get("geometry_mismatches", 0) != 0)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_multi_rank(args)
    import torch
    import torch.distributed as dist
    from dagsfm_amd import capi, sharding, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    dev_index = 0 if args.oversubscribe else local_rank
    if dev_index >= n_dev:
        raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible" % (rank, dev_index, n_dev))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm_dev = dev
    # The contexts come BEFORE the process group: a context creates its streams at once (its own + the second verification lane's) and
    # the HIP runtime hands out hardware queues in creation order; behind RCCL's streams the two lanes shared a queue and the
    # verification lost the overlap of its lanes (322 vs 288 ms per step).  --ctx-after-pg: the old order (A/B).
    n_ctx = max(1, args.contexts)
    ctxs = [] if args.ctx_after_pg else [capi.Context(dev_index) for _ in range(n_ctx)]
    if world > 1:
        if args.oversubscribe:  # several ranks on one GPU: RCCL refuses that, gloo moves host copies
            dist.init_process_group("gloo")
            comm_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == world
    elif args.force_collectives:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    force = {"": None, "auto": True}.get(args.force_collectives, args.force_collectives)

    verify = not args.no_verify
    calibrated = not args.uncalibrated
    scene = synthetic.Scene(args.images, args.feats, seed=args.seed, outlier_frac=args.outlier_frac, planar=args.planar)
    if args.pairs == "exhaustive":
        pairs = synthetic.exhaustive_pairs(args.images)
        pairs_desc = "exhaustive"
    elif args.pairs.startswith("knn:"):
        pairs = synthetic.knn_pairs(scene, args.images, int(args.pairs[4:]), args.seed)
        pairs_desc = "kNN candidate graph, %s neighbours/image, id1<id2" % args.pairs[4:]
    else:
        raise SystemExit("--pairs must be exhaustive or knn:K")
    n_full = len(pairs)
    if args.shard_of > 1:
        if not 0 <= args.shard_index < args.shard_of:
            raise SystemExit("--shard-index must be in [0, --shard-of)")
        if args.cut == "interleaved":
            pairs = pairs[sharding.interleaved_parts(len(pairs), args.shard_of)[args.shard_index]]
        else:
            pairs = sharding.shard(pairs, args.shard_index, args.shard_of)
    if args.max_pairs and len(pairs) > args.max_pairs:
        pairs = pairs[:args.max_pairs]
    # only the images the (sharded / truncated) list touches are generated and made resident
    used = np.unique(pairs) if (args.shard_of > 1 or args.max_pairs) else np.arange(args.images)
    remap = np.full(args.images, -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    images = [scene.image(int(i)) for i in used]
    pairs = remap[pairs.astype(np.int64)].astype(np.uint32)
    # cut by cost (N1 * N2 + a per-pair term), like the C++ shim cuts between the devices of gpu_index
    costs = sharding.pair_costs(pairs, [len(im[0]) for im in images])
    order = None
    if args.cut == "interleaved" and world > 1:
        parts = sharding.interleaved_parts(len(pairs), world, costs)
        bounds, order = sharding.parts_bounds_and_order(parts)
        my_pairs = pairs[parts[rank]]
    else:
        bounds = sharding.shard_bounds(len(pairs), world, costs)
        my_pairs = pairs[bounds[rank]:bounds[rank + 1]]

    if not ctxs:
        ctxs = [capi.Context(dev_index) for _ in range(n_ctx)]
    ctx = ctxs[0]
    if args.memory_budget_gib > 0:
        for c in ctxs:
            c.set_memory_budget(int(args.memory_budget_gib * (1 << 30)))
    info = ctx.device_info()
    cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, calibrated)
            for _ in range(len(images))]
    handover_s = []
    for c in ctxs:
        th = time.perf_counter()
        c.set_images([im[0] for im in images], [im[1] for im in images], cams)  # (synchronous: the images are in HBM when it returns)
        handover_s.append(time.perf_counter() - th)
    # (the first call of a context also allocates the resident buffers and loads the code object: a second one shows the transfer alone)
    th = time.perf_counter()
    ctxs[0].set_images([im[0] for im in images], [im[1] for im in images], cams)
    handover_repeat_s = time.perf_counter() - th
    cbounds = sharding.shard_bounds(len(my_pairs), n_ctx)
    cparts = [my_pairs[cbounds[k]:cbounds[k + 1]] for k in range(n_ctx)]
    opts = capi.default_match_options()
    topts = capi.default_two_view_options()
    if args.fixed_trials:
        topts = capi.default_two_view_options(min_num_trials=args.fixed_trials, max_num_trials=args.fixed_trials,
                                              confidence=0.999999, min_inlier_ratio=0.01)
    user_seed = 0
    source = sharding.CtxSource(ctx, len(my_pairs), dev) if n_ctx == 1 else sharding.MultiCtxSource(ctxs, [len(p) for p in cparts], dev)
    if comm_dev.type == "cpu":
        class HostSource:  # gloo debug path: the same fetches, staged through the host
            def __getattr__(self, name):
                f = getattr(source, name)
                return lambda *a: f(*a).cpu()
        gsource = HostSource()
    else:
        gsource = source

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import threading
    match_turn = threading.Lock()

    def run_part(k):
        # several contexts: the matching calls take turns (they all want the matrix pipe), each context's verification
        # (FP64 VALU, latency chains) then runs next to the following context's matching -- the reference's matcher and
        # verifier thread pools overlap the same way (/root/reference/src/feature/matching.cc:640-674)
        if args.no_match_lock:
            ctxs[k].match_pairs(cparts[k], opts)
        else:
            with match_turn:
                ctxs[k].match_pairs(cparts[k], opts)
        if verify:
            ctxs[k].verify_pairs(topts, user_seed=user_seed, stage_filter=True)

    gather_s = [0.0]  # fetch of this rank's results into torch tensors + the exchange (RCCL all-gather / broadcast)

    def step():
        if n_ctx == 1:
            run_part(0)
        else:
            th = [threading.Thread(target=run_part, args=(k,)) for k in range(n_ctx)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        tg = time.perf_counter()
        g = sharding.gather_match_graph(dist, gsource, rank, world, bounds, verify, force_collectives=force, order=order)
        torch.cuda.synchronize()  # inside the timed region on purpose: a step ends when the assembled graph is complete in HBM
        gather_s[0] += time.perf_counter() - tg
        return g

    for _ in range(args.warmup):
        step()
    k1_ms, k1_launches, kv_ms, k1b_ms, k1g_ms, k1t_ms = 0.0, 0, 0.0, 0.0, 0.0, 0.0
    graph = None
    barrier()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph = step()
        for c in ctxs:  # host-side reads of HIP-event times already taken inside the library
            ms, nl = c.match_kernel_time()
            k1_ms += ms
            k1_launches += nl
            k1b_ms += c.match_resolve_time()
            k1g_ms += c.match_gather_time()
            k1t_ms += c.match_tail_time()
            if verify:
                kv_ms += c.verify_kernel_time()
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        # what every rank did, so that a scaling curve explains itself: the step time of the slowest rank is the job's (max below),
        # the spread between min and max is the imbalance of the cut, match / verify say which stage carries it
        mine = {"ms_per_step": 1e3 * dt / args.steps, "match_ms": (k1_ms + k1b_ms + k1g_ms + k1t_ms) / args.steps,
                "verify_ms": kv_ms / args.steps, "exchange_ms": 1e3 * gather_s[0] / args.steps, "pairs": int(len(my_pairs))}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        per_rank = {k: [min(r[k] for r in allr), sum(r[k] for r in allr) / world, max(r[k] for r in allr)] for k in mine}
        per_rank["order"] = "min, mean, max over the ranks"
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    failures, exit_code = [], 0
    # ---- statistics of the assembled graph (outside the timed region)
    n_pairs = len(pairs)
    assert graph.match_counts.shape[0] == n_pairs, "the assembled match graph must cover the whole pair list"
    res = dict(matches=int(graph.matches.shape[0]), inliers=0, models=0, verified=0, score_flops=0.0)
    if verify:
        tvb = ctypes.sizeof(capi.TwoViewGeometry)
        rec = graph.tvg
        head = rec[:, :16].contiguous().view(torch.int32)           # config, num_inliers, num_matches, reserved
        tail = rec[:, tvb - 16:].contiguous().view(torch.int32)     # num_models[4]
        res["verified"] = int((head[:, 0] > 1).sum().item())
        res["models"] = int(tail.sum().item())
        # algorithmic FP64 flops of the inlier scoring: per (model, correspondence) 33 Sampson (E, F),
        # 20 transfer (H), 5 translation (watermark)
        w = torch.tensor([33.0, 33.0, 20.0, 5.0], dtype=torch.float64, device=rec.device)
        res["score_flops"] = float(((tail.to(torch.float64) * w).sum(dim=1) * head[:, 2].to(torch.float64)).sum().item())
        res["inliers"] = int(graph.inlier_matches.shape[0])

    if rank == 0 and args.dump_graph:
        np.savez(args.dump_graph, match_counts=graph.match_counts.cpu().numpy(), matches=graph.matches.cpu().numpy(),
                 **({"tvg": graph.tvg.cpu().numpy(), "inlier_counts": graph.inlier_counts.cpu().numpy(),
                     "inlier_matches": graph.inlier_matches.cpu().numpy()} if verify else {}))
    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n_pairs * args.steps / dt
        # ---- peaks from what the device reports (hipDeviceProp_t), the guide's figures as a sanity bound
        cus, clk = info.compute_units, info.clock_khz * 1e3
        int8_peak = cus * clk * 4 * 2048.0   # per CU and clock: 4 SIMDs x (32x32x32 i8 MFMA = 65 536 ops / 32 cycles)
        fp64_peak = cus * clk * 4 * 32.0     # 4 SIMD-32 x 16 lanes-equivalent of f64 FMA x 2 flops = 128 flop/clk/CU
        hbm_peak = info.memory_clock_khz * 1e3 * 2.0 * info.memory_bus_bits / 8.0
        peaks_note = "from hipDeviceProp: %d CUs x %.0f MHz; HBM %d-bit x %.0f MHz DDR" % (
            cus, clk / 1e6, info.memory_bus_bits, info.memory_clock_khz / 1e3)
        if not (0.5 * GUIDE_INT8_DENSE <= int8_peak <= 1.2 * GUIDE_INT8_DENSE):
            peaks_note += "; derived int8 peak %.3g outside the guide's range -> guide value used" % int8_peak
            int8_peak, fp64_peak = GUIDE_INT8_DENSE, GUIDE_FP64_VECTOR
        if not (0.5 * GUIDE_HBM <= hbm_peak <= 1.2 * GUIDE_HBM):
            hbm_peak = GUIDE_HBM
        # ---- roofline of the dominant kernel (k1_best_rows, both passes) on this rank:
        # algorithmic ops = 2*128*N1*N2 per pair x pairs per launch
        ops_per_pair = 2.0 * 128.0 * args.feats * args.feats
        launches = max(k1_launches, 1)
        pass1_s = 1e-3 * k1_ms / launches
        pass2_s = 1e-3 * k1g_ms / launches
        pairs_per_launch = len(my_pairs) * args.steps / launches
        achieved = ops_per_pair * pairs_per_launch / (pass1_s + pass2_s) if pass1_s > 0 else 0.0
        fp = {}
        if world == 1 and args.pairs == "exhaustive" and args.shard_of == 1 and not args.max_pairs and not args.fixed_trials \
                and abs(args.outlier_frac - 0.2) < 1e-12 and not args.planar:
            fp = profile_figures(ROOT, args.images, args.feats, n_pairs, verify and calibrated)
        traffic = fp.get("k1", {}).get("hbm_bytes_per_launch")
        traffic_file = fp.get("k1", {}).get("file")
        fam = ("calibrated: E+F+H + relative pose" if calibrated else "uncalibrated: F+H")
        if args.fixed_trials:
            fam += ", fixed %d trials/family" % args.fixed_trials
        shard_note = ""
        if args.shard_of > 1 or args.max_pairs:
            shard_note = " [shard %d of %d of %d pairs%s]" % (args.shard_index + 1, args.shard_of, n_full, ", truncated" if args.max_pairs else "")
        out = {
            "metric": ("verified image-pairs/sec (+ RANSAC hypotheses/sec) at %d feats/image" % args.feats) if verify else
                      "matched image-pairs/sec at %d feats/image (matching only, --no-verify)" % args.feats,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 (int8 MFMA, int32 accumulate)" if not verify else "u8 matching (int8 MFMA) + f64 verification",
            "data": "synthetic",
            "config": {"workload": "%d images x %d feats, %s (%d pairs)%s, %s" % (
                args.images, args.feats, pairs_desc, n_pairs, shard_note,
                (("match + two-view LO-RANSAC (%s)" % fam) if verify else "match only") + (", planar scene" if args.planar else "")),
                "pairs": n_pairs, **({"images_resident": len(images)} if len(images) != args.images else {}), "total_matches": res["matches"],
                "total_inlier_matches": res["inliers"], "pairs_with_geometry": res["verified"],
                "hypotheses_per_step": res["models"],
                "putative_match_inlier_ratio": round((1.0 - args.outlier_frac) ** 2, 4),
                **({"contexts_per_gpu": n_ctx} if n_ctx != 1 else {}),
                "parallelism": "pair-sharded x%d (%s cut) + %s all-gather of the match graph" % (
                    world, args.cut, "gloo (debug, oversubscribed)" if args.oversubscribe else "RCCL")},
            "hypotheses_per_s": res["models"] * args.steps / dt if verify else None,
            "device": {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": cus, "clock_mhz": clk / 1e6,
                       "hbm_gb": info.total_memory / 2 ** 30, "ranks_seen_by_process_group": world},
            "kernel_ms_per_step": {"k1_best_rows<pass 1>": k1_ms / args.steps, "k1_best_rows<gathered pass 2>": k1g_ms / args.steps,
                                   "k1_resolve_index<pass 1>": k1b_ms / args.steps,
                                   "pass-2 k1_resolve_index + compaction": k1t_ms / args.steps, "k_verify_pairs": kv_ms / args.steps,
                                   "exchange": 0.0},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": int8_peak / 1e12,
                         "unit": "TFLOP/s", "frac": achieved / int8_peak, "traffic": traffic,
                         "traffic_file": traffic_file,
                         "traffic_note": "HBM bytes per launch of both passes, rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-read "
                                         "correction) + WRITE_SIZE in separate passes, read from the committed collection "
                                         "traffic_file (an earlier run of this workload); null when not collected for it",
                         "kernel": "k1_best_rows (pass 1 + gathered pass 2)",
                         "avg_launch_ms": 1e3 * (pass1_s + pass2_s), "avg_launch_ms_pass1": 1e3 * pass1_s,
                         "avg_launch_ms_pass2": 1e3 * pass2_s, "launches": k1_launches, "peak_source": peaks_note,
                         "note": "int8 ops (2 per MAC) counted as flops; algorithmic = ONE 2*128*N1*N2 distance matrix per pair "
                                 "(SURVEY.md 8d) over the HIP-event time of BOTH k1_best_rows launches; pass 2 recomputes only "
                                 "the rows matches12 points at (~7 % of the matrix at this shape)"},
        }
        # ---- the exchange: fetch of the rank's results into torch tensors + all-gather / broadcast of the match graph
        result_bytes = 16 * n_pairs + 8 * res["matches"] + (ctypes.sizeof(capi.TwoViewGeometry) * n_pairs + 8 * res["inliers"] if verify else 0)
        out["exchange"] = {"gather_ms_per_step": 1e3 * gather_s[0] / args.steps, "bytes_per_step": result_bytes,
                           "backend": (dist.get_backend() if dist.is_initialized() else "none"),
                           "forced_on_one_rank": bool(world == 1 and force),
                           "note": "device-to-device fetch through the C-ABI getters + the collectives of sharding.gather_match_graph "
                                   "(rank 0's wall time, inside the timed region)"}
        out["kernel_ms_per_step"]["exchange"] = out["exchange"]["gather_ms_per_step"]
        # the boundary hands over HOST buffers (dsm_set_images): outside the timed region, once per job.  Its measured time and
        # the rate a job would see that paid it before EVERY step (never `value`)
        image_bytes = sum(im[0].nbytes + (im[1].nbytes if im[1] is not None else 0) for im in images)
        out["host_handover"] = {"ms": 1e3 * min(handover_s), "ms_repeat": 1e3 * handover_repeat_s, "bytes": image_bytes,
                                "value_if_paid_every_step": n_pairs / (ms_per_step / 1e3 + min(handover_s)) if world == 1 else None}
        if pass1_s > 0 and res["matches"] >= 0:
            # what the matrix pipe executed: pass 1 = the whole matrix; pass 2 = gathered rows in 128-row wave units
            out["roofline"]["frac_pass1_only"] = ops_per_pair * pairs_per_launch / pass1_s / int8_peak
        if verify and kv_ms > 0:
            ach = res["score_flops"] / (1e-3 * kv_ms / args.steps) / max(world, 1)
            out["roofline_verify"] = {"bound": "fp64-valu", "achieved": ach / 1e12, "peak": fp64_peak / 1e12, "unit": "TFLOP/s",
                                      "frac": ach / fp64_peak, "traffic": None,
                                      "note": "algorithmic inlier-scoring flops only (33 / 20 / 5 per model x correspondence), "
                                              "per GPU, over the HIP-event time of all verification kernels"}
        if fp:
            # counter figures of an EARLIER run of this workload (files named inside); instruction counts are per launch and
            # do not depend on the clock, so executed work / this run's launch time is printed next to them
            if fp.get("k1", {}).get("mfma_i8_insts_per_launch") and pass1_s > 0:
                fp["k1"]["executed_frac_at_this_runs_time"] = fp["k1"]["mfma_i8_insts_per_launch"] * 65536.0 / (pass1_s + pass2_s) / int8_peak
            out["from_profiles"] = fp
        if per_rank is not None:
            out["per_rank"] = per_rank
        if args.memory_budget_gib > 0:
            res_b, scr_b = ctx.memory_footprint()
            out["memory"] = {"budget_gib": args.memory_budget_gib, "scratch_gib": scr_b / 2 ** 30, "resident_gib": res_b / 2 ** 30}
        # ---- side measurements after the timed region (never part of `value`), each held against the oracle on a bounded sample of
        # its own pairs (`parity_sample`): what is timed is what is checked
        side_ok = (world == 1 and verify and args.shard_of == 1 and not args.max_pairs and args.pairs == "exhaustive" and not args.fixed_trials
                   and not args.uncalibrated)
        cores = min(host_cores(), 256)
        orc = None
        if args.cpu_seconds > 0 and world == 1:
            from tests import oracle_lib
            orc = oracle_lib.load()

        def ctx_view(n, with_geometry):
            src = sharding.CtxSource(ctx, n, dev)
            mo = src.match_offsets()
            m = src.matches(int(mo[-1].item()))
            if not with_geometry:
                return GraphView(mo[1:] - mo[:-1], m)
            io = src.inlier_offsets()
            return GraphView(mo[1:] - mo[:-1], m, src.two_view_geometries(), io[1:] - io[:-1], src.inlier_matches(int(io[-1].item())))

        def side_run(ims, pr, cm, n_timed, with_geometry, oracle_s, every_pair=False):
            """one warm-up + n_timed passes of match (+ verify) over `pr` on the context; then the oracle on a sample of the same pairs"""
            ctx.set_images([im[0] for im in ims], [im[1] for im in ims] if with_geometry else None, cm if with_geometry else None)
            ts_, kv_ = [], 0.0
            for it in range(1 + n_timed):
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                ctx.match_pairs(pr, opts)
                if with_geometry:
                    ctx.verify_pairs(topts, user_seed=user_seed, stage_filter=True)
                ctx.sync()
                if it:
                    ts_.append(time.perf_counter() - t_a)
                    kv_ += ctx.verify_kernel_time() if with_geometry else 0.0
            r = {"pairs": int(len(pr)), "pairs_per_s": len(pr) * len(ts_) / sum(ts_), "ms_per_step": 1e3 * sum(ts_) / len(ts_)}
            if with_geometry:
                r["verify_us_per_pair"] = 1e3 * kv_ / len(ts_) / len(pr)
            if orc is not None and oracle_s > 0:
                keep = {}
                cb = cpu_baseline(orc, "-O3", "-O3", ims, pr, oracle_s, with_geometry, cm, topts, user_seed, cores, keep=keep, every_pair=every_pair)
                r["cpu_pairs_per_s"] = cb["value"]
                r["cpu_sample_note"] = cb["sample"]
                r["parity_sample"] = parity_sample(keep, ctx_view(len(pr), with_geometry), with_geometry, int(topts.min_num_inliers))
                if parity_failed(r["parity_sample"]):
                    failures.append(r["parity_sample"])
                if "pose_max_rel" in r["parity_sample"]:  # (the headline's parity_sample keeps it in the line)
                    r["parity_sample"]["side_pose_max_rel"] = r["parity_sample"].pop("pose_max_rel")
            return r

        extra = {}
        if side_ok and not args.no_second_regime and args.images >= 150 and args.outlier_frac < 0.5:
            # the same pipeline where real collections live: half of every image's features are not observations
            # of the scene, a putative match is right with 0.25 instead of 0.64, RANSAC needs ~13x the trials.  150 images
            try:
                n2 = 150
                scene2 = synthetic.Scene(n2, args.feats, seed=args.seed, outlier_frac=0.5)
                r = side_run([scene2.image(i) for i in range(n2)], synthetic.exhaustive_pairs(n2), cams[:n2], 2, True, min(2.0, args.cpu_seconds))
                r["workload"] = "%d x %d, inlier ratio 0.25" % (n2, args.feats)
                extra["low_inlier_regime"] = r
            except Exception as e:  # a side measurement must never cost the headline line
                extra["low_inlier_regime"] = {"error": repr(e)}
        if side_ok and not args.no_extra_configs and args.images == 500:
            # BASELINE configs[1]'s other half ("F-path and E-path both reported"): the same 124 750 pairs with
            # cameras without a focal prior -- EstimateUncalibrated, F + H (/root/reference/src/estimators/two_view_geometry.cc:427-489)
            try:
                cams_u = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, False)
                          for _ in range(len(images))]
                r = side_run(images, pairs, cams_u, 2, True, min(3.0, args.cpu_seconds))
                r["workload"] = "%d x %d, uncalibrated: F+H" % (args.images, args.feats)
                extra["uncalibrated"] = r
            except Exception as e:
                extra["uncalibrated"] = {"error": repr(e)}
            # BASELINE configs[0] on the GPU: 50 images x 1 024 features, 1 225 pairs, F + H; small enough for the
            # oracle to check EVERY pair
            try:
                n1 = 50
                scene1 = synthetic.Scene(n1, 1024, seed=42)
                cams1 = [capi.simple_pinhole(scene1.focal, scene1.width / 2.0, scene1.height / 2.0, scene1.width, scene1.height, False)
                         for _ in range(n1)]
                r = side_run([scene1.image(i) for i in range(n1)], synthetic.exhaustive_pairs(n1), cams1, 10, True,
                             min(60.0, 6.0 * args.cpu_seconds), every_pair=True)
                r["workload"] = "50 x 1024, uncalibrated: F+H (BASELINE configs[0]), every pair checked"
                extra["config1"] = r
            except Exception as e:
                extra["config1"] = {"error": repr(e)}
        if side_ok and not args.no_config3 and args.images == 500:
            # BASELINE configs[2] (missing 6): 2 000 images x the same feature count, exhaustive, matching only, with the CPU
            # matcher of oracle/ on the host's cores beside it -- the shape north_star's ">= 10x the host-CPU baseline" is stated on
            try:
                n3 = 2000
                scene3 = synthetic.Scene(n3, args.feats, seed=args.seed, outlier_frac=args.outlier_frac)
                im3 = [(scene3.image(i)[0], None) for i in range(n3)]
                r = side_run(im3, synthetic.exhaustive_pairs(n3), None, 1, False, min(5.0, args.cpu_seconds))
                r["k1_pass1_ms"] = ctx.match_kernel_time()[0]
                r["workload"] = "%d x %d, match only (BASELINE configs[2])" % (n3, args.feats)
                if "cpu_pairs_per_s" in r:
                    r["gpu_over_cpu"] = r["pairs_per_s"] / max(r["cpu_pairs_per_s"], 1e-9)
                extra["config3_match_only"] = r
                del im3
            except Exception as e:
                extra["config3_match_only"] = {"error": repr(e)}
        if extra:
            out["extra"] = extra
        # ---- the CPU baseline on a bounded sample of the headline workload -- and the graph of the LAST TIMED STEP held against the
        # oracle's results for exactly those pairs
        if orc is not None:
            share = args.cpu_seconds * 0.6
            keep = {}
            out["cpu_baseline"] = cpu_baseline(orc, "built -O3 without -march (CMake Release, like the reference)",
                                               "-O3", images, pairs, share, verify, cams, topts, user_seed, cores, keep=keep)
            native = oracle_lib.load_native()
            if native is not None:
                out["cpu_baseline_native"] = cpu_baseline(native, "built -O3 -march=native on this host (labelled second baseline, SURVEY 8d)",
                                                          "-O3 -march=native", images, pairs, args.cpu_seconds - share, verify, cams,
                                                          topts, user_seed, cores, keep=keep)
            out["parity_sample"] = parity_sample(keep, GraphView(graph=graph), verify, int(topts.min_num_inliers))
            if parity_failed(out["parity_sample"]):
                failures.append(out["parity_sample"])
        result_line = format_line(out, args.dump_line)
    # RCCL prints a version banner through C stdio when the first communicator comes up; redirected to a file or a pipe it sits
    # in the C buffer until exit -- i.e. it would land BEHIND the result line.  Every rank flushes the C streams now, the ranks
    # meet, and only then does rank 0 print: the JSON line is the last thing on stdout.
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(result_line, flush=True)
        if failures:
            exit_code = 3
            print("bench.py: the device's results differ from the oracle's on a sampled pair: %r" % failures, file=sys.stderr)
    if dist.is_initialized():
        dist.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
