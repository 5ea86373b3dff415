#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: verified image-pairs/s at 4 096 feats/image.

One "step" = one pass of the hot path (brute-force matching + two-view verification) over the
whole exhaustive pair list of the workload (BASELINE.json configs[1]: 500 images x 4 096
features => 124 750 pairs), inputs already resident in HBM.  With N GPUs the pair list is
block-partitioned over the ranks (strong scaling), every rank runs the same kernels on its
share, and the per-pair results are all-gathered with RCCL so that every rank holds the full
match graph.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement").
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_MFMA_DENSE_PEAK = 5.0e15  # ops/s, MI355X dense (MI355X_MICROARCH.md: ~5 PF dense 8-bit; 4.40 P measured)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-verify", action="store_true", help="matching only (BASELINE config 3 style)")
    ap.add_argument("--uncalibrated", action="store_true",
                    help="cameras without focal prior: F + H path (EstimateUncalibrated) instead of E + F + H + pose")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def cpu_baseline(scene_images, pairs, budget_s, verify, cams=None, opts=None, user_seed=0):
    """Times the CPU oracle (the reference algorithm restated, oracle/) on a bounded sample of the same
    workload: worker threads pull evenly spaced pairs of the list until `budget_s` seconds have passed,
    one thread per usable host core like the reference's matcher/verifier thread pools
    (/root/reference/src/feature/matching.cc:640-674)."""
    import threading
    from tests import oracle_lib
    from dagsfm_amd import capi
    orc = oracle_lib.load()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # each worker holds an N1 x N2 int32 distance matrix (64 MiB at 4096 features) like the reference does;
    # beyond ~64 threads the host's memory system, not its cores, limits this path
    cores = min(cores, 64)
    kps = [im[1].astype(np.float64) for im in scene_images]
    order = np.linspace(0, len(pairs) - 1, min(len(pairs), 65536)).astype(np.int64)
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
            nm = 0
            if verify:
                tv, _ = orc.estimate_two_view_geometry(cams[i], kps[i], cams[j], kps[j], m, opts, capi.pair_seed(i, j, user_seed))
                nm = sum(tv.num_models)
            with lock:
                state["pairs"] += 1
                state["models"] += nm

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker) for _ in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": state["pairs"] / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "hypotheses_per_s": state["models"] / dt,
            "sample": "%d of %d pairs (%s) in %.1f s on %d threads; oracle/ = the reference CPU path restated "
                      "(MatchSiftFeaturesCPU + TwoViewGeometry::Estimate), built -O3 without -march (CMake Release, like the reference)"
                      % (state["pairs"], len(pairs), "match only" if not verify else "match + verify", dt, cores)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from dagsfm_amd import capi, sharding, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    verify = not args.no_verify
    calibrated = not args.uncalibrated
    scene = synthetic.Scene(args.images, args.feats, seed=args.seed)
    images = [scene.image(i) for i in range(args.images)]
    pairs = synthetic.exhaustive_pairs(args.images)
    # strong scaling: contiguous block of the pair list per rank
    bounds = sharding.shard_bounds(len(pairs), world)
    my_pairs = sharding.shard(pairs, rank, world)

    ctx = capi.Context(local_rank)
    cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, calibrated)
            for _ in range(args.images)]
    ctx.set_images([im[0] for im in images], [im[1] for im in images], cams)
    opts = capi.default_match_options()
    topts = capi.default_two_view_options()
    user_seed = 0
    TVG_BYTES = ctypes.sizeof(capi.TwoViewGeometry)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_var(fetch, total):
        """All-gather of a variable-length [total, 2] int32 array produced by `fetch(ptr, capacity)`."""
        if world == 1:
            return total
        mine = torch.zeros((max(total, 1), 2), dtype=torch.int32, device=dev)
        if total:
            fetch(mine.data_ptr(), total)
        sizes, _ = sharding.all_gather_ragged(dist, mine[:total], world)
        return int(sizes.sum())

    def gather_results():
        """All-gather of the per-pair match graph over RCCL: matches and, when verifying, the
        TwoViewGeometry records + inlier matches."""
        L = capi.lib()
        offs = torch.empty(len(my_pairs) + 1, dtype=torch.int64, device=dev)
        L.dsm_get_matches(ctx._h, offs.data_ptr(), None, 0)
        total = int(offs[-1].item())
        n_matches = gather_var(lambda ptr, cap: L.dsm_get_matches(ctx._h, None, ptr, cap), total)
        n_inl, n_models, n_ok, score_flops = 0, 0, 0, 0.0
        if verify:
            maxp = int(np.diff(bounds).max())
            tv = torch.zeros((len(my_pairs), TVG_BYTES), dtype=torch.uint8, device=dev)
            L.dsm_get_two_view_geometries(ctx._h, tv.data_ptr())
            rec = sharding.all_gather_fixed(dist, tv, maxp, world) if world > 1 else tv
            head = rec[:, :16].contiguous().view(torch.int32)              # config, num_inliers, num_matches, reserved
            tail = rec[:, TVG_BYTES - 16:].contiguous().view(torch.int32)  # num_models[4]
            n_ok = int((head[:, 0] > 1).sum().item())
            n_models = int(tail.sum().item())
            # algorithmic FP64 flops of the inlier scoring: per (model, correspondence)
            # 33 Sampson (E, F), 20 transfer (H), 5 translation (watermark)
            w = torch.tensor([33.0, 33.0, 20.0, 5.0], dtype=torch.float64, device=dev)
            score_flops = float((tail.to(torch.float64) @ w * head[:, 2].to(torch.float64)).sum().item())
            ioffs = torch.empty(len(my_pairs) + 1, dtype=torch.int64, device=dev)
            L.dsm_get_inlier_matches(ctx._h, ioffs.data_ptr(), None, 0)
            itotal = int(ioffs[-1].item())
            n_inl = gather_var(lambda ptr, cap: L.dsm_get_inlier_matches(ctx._h, None, ptr, cap), itotal)
        return dict(matches=n_matches, inliers=n_inl, models=n_models, verified=n_ok, score_flops=score_flops)

    def step():
        ctx.match_pairs(my_pairs, opts)
        if verify:
            ctx.verify_pairs(topts, user_seed=user_seed, stage_filter=True)
        return gather_results()

    for _ in range(args.warmup):
        step()
    k1_ms, k1_launches, kv_ms, k1b_ms = 0.0, 0, 0.0, 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        ms, nl = ctx.match_kernel_time()
        k1_ms += ms
        k1_launches += nl
        k1b_ms += ctx.match_resolve_time()
        if verify:
            kv_ms += ctx.verify_kernel_time()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        n_pairs = len(pairs)
        ms_per_step = 1e3 * dt / args.steps
        value = n_pairs * args.steps / dt
        # roofline of the dominant kernel (k1_best_rows) on this rank:
        # algorithmic ops = 2*128*N1*N2 per pair x pairs per launch
        ops_per_pair = 2.0 * 128.0 * args.feats * args.feats
        avg_launch_s = 1e-3 * k1_ms / max(k1_launches, 1)
        pairs_per_launch = len(my_pairs) * args.steps / max(k1_launches, 1)
        achieved = ops_per_pair * pairs_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
        traffic = None
        try:  # HBM bytes per K1 launch from the committed PMC collection (tools/collect_pmc.py), same workload only
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_k1_pmc.json")))
            if pmc.get("images") == args.images and pmc.get("feats") == args.feats and world == 1:
                traffic = pmc.get("k1_traffic_bytes_per_launch")
        except Exception:
            traffic = None
        out = {
            "metric": ("verified image-pairs/sec (+ RANSAC hypotheses/sec) at %d feats/image" % args.feats) if verify else
                      "matched image-pairs/sec at %d feats/image (matching only, --no-verify)" % args.feats,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 (int8 MFMA, int32 accumulate)" if not verify else "u8 matching (int8 MFMA) + f64 verification",
            "data": "synthetic",
            "config": {"workload": "%d images x %d feats exhaustive (%d pairs), %s" % (
                args.images, args.feats, n_pairs,
                ("match + two-view LO-RANSAC (%s)" % ("calibrated: E+F+H + relative pose" if calibrated else "uncalibrated: F+H"))
                if verify else "match only"),
                "pairs": n_pairs, "total_matches": res["matches"], "total_inlier_matches": res["inliers"],
                "pairs_with_geometry": res["verified"], "hypotheses_per_step": res["models"],
                "parallelism": "pair-sharded x%d + RCCL all-gather" % world},
            "hypotheses_per_s": res["models"] * args.steps / dt if verify else None,
            "kernel_ms_per_step": {"k1_best_rows": k1_ms / args.steps, "k1_resolve_index": k1b_ms / args.steps,
                                   "k_verify_pairs": kv_ms / args.steps},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": INT8_MFMA_DENSE_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": achieved / INT8_MFMA_DENSE_PEAK, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-read correction) + "
                                         "WRITE_SIZE in separate passes (profiles/r01_k1_pmc.json); null when not collected for this workload",
                         "kernel": "k1_best_rows", "avg_launch_ms": 1e3 * avg_launch_s, "launches": k1_launches,
                         "executed_frac": 2.0 * achieved / INT8_MFMA_DENSE_PEAK,
                         "note": "int8 ops (2 per MAC) counted as flops; algorithmic = ONE 2*128*N1*N2 distance matrix per pair "
                                 "(SURVEY.md 8d).  The kernel issues twice that (one directed pass per direction of the "
                                 "cross-check, each with its own fused top-2): executed_frac is the matrix-pipe view"},
        }
        if verify and kv_ms > 0:
            # second roofline, verification: algorithmic scoring flops of the step / device time of the verification
            # kernels (solvers, local optimisation and the sequential replay are extra work on top of it)
            fp64_peak = 78.6e12  # MI355X vector FP64 (MI355X_MICROARCH.md)
            ach = res["score_flops"] / (1e-3 * kv_ms / args.steps) / max(world, 1)
            out["roofline_verify"] = {"bound": "fp64-valu", "achieved": ach / 1e12, "peak": fp64_peak / 1e12, "unit": "TFLOP/s",
                                      "frac": ach / fp64_peak, "traffic": None,
                                      "note": "algorithmic inlier-scoring flops only (33 / 20 / 5 per model x correspondence), "
                                              "per GPU, over the HIP-event time of all verification kernels"}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(images, pairs, args.cpu_seconds, verify, cams, topts, user_seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
