#!/usr/bin/env python3
"""Multi-GPU readiness without the node: every shard of an S-way split of a pair list, run one
after the other on ONE GPU with all images resident, exactly as rank r of an S-GPU job would run it (same library
calls, same cost-aware cut as bench.py).  Reports per-shard step time, max / mean (the imbalance a barrier would
see), the whole list's time on the same GPU and the strong scaling that projects to:

    projected = T_whole / (max shard + all-gather estimate)

The all-gather is NOT measured (no second GPU): it is priced at result bytes x (S-1)/S over one xGMI link direction
(153 GB/s nameplate, MI355X_MICROARCH.md) at an assumed 60 % efficiency and labelled as an estimate.

    python tools/shard_sweep.py --images 500 --feats 4096 [--pairs knn:200] [--shards 8] [--steps 2]"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import capi, sharding, synthetic  # noqa: E402

_SCENE = None


def _init(args):
    global _SCENE
    _SCENE = synthetic.Scene(*args[:2], seed=args[2], outlier_frac=args[3])


def _image(i):
    return _SCENE.image(i)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--pairs", default="exhaustive")
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--outlier-frac", type=float, default=0.2)
    ap.add_argument("--max-pairs", type=int, default=0, help="truncate the list (bounded runs)")
    ap.add_argument("--skip-whole", action="store_true", help="do not time the unsharded list (too long for the big configs)")
    ap.add_argument("--cut", default="both", choices=["contiguous", "interleaved", "both"],
                    help="the cut(s) to sweep: one contiguous cost-balanced stretch per rank / blocks of 256 pairs dealt out (sharding.interleaved_parts)")
    a = ap.parse_args()
    scene_args = (a.images, a.feats, a.seed, a.outlier_frac)
    _init(scene_args)
    if a.pairs == "exhaustive":
        pairs = synthetic.exhaustive_pairs(a.images)
    else:
        pairs = synthetic.knn_pairs(_SCENE, a.images, int(a.pairs.split(":")[1]), a.seed)
    if a.max_pairs:
        pairs = pairs[:a.max_pairs]
    t0 = time.perf_counter()
    with Pool(min(64, os.cpu_count() or 1), initializer=_init, initargs=(scene_args,)) as pool:
        images = pool.map(_image, range(a.images), chunksize=8)
    t_gen = time.perf_counter() - t0
    ctx = capi.Context(0)
    cams = [capi.simple_pinhole(_SCENE.focal, _SCENE.width / 2.0, _SCENE.height / 2.0, _SCENE.width, _SCENE.height, 1) for _ in range(a.images)]
    ctx.set_images([im[0] for im in images], [im[1] for im in images], cams)
    mo, to = capi.default_match_options(), capi.default_two_view_options()
    costs = sharding.pair_costs(pairs, [len(im[0]) for im in images])
    bounds = sharding.shard_bounds(len(pairs), a.shards, costs)
    cuts = {"contiguous": [pairs[bounds[r]:bounds[r + 1]] for r in range(a.shards)],
            "interleaved": [pairs[ix] for ix in sharding.interleaved_parts(len(pairs), a.shards, costs)]}
    if a.cut != "both":
        cuts = {a.cut: cuts[a.cut]}

    def run(pl):
        ctx.match_pairs(pl, mo)
        ctx.verify_pairs(to, user_seed=0, stage_filter=True)
        ctx.sync()

    def timed(pl):
        run(pl)  # warm-up: scratch allocation, tables
        t = time.perf_counter()
        for _ in range(a.steps):
            run(pl)
        wall = (time.perf_counter() - t) / a.steps * 1e3
        k1, n = ctx.match_kernel_time()
        offs, _ = ctx.matches()
        ioffs, _ = ctx.inlier_matches()
        result_bytes = len(pl) * (8 + 328) + 8 * (int(offs[-1]) + int(ioffs[-1]))
        return wall, k1 / max(n, 1) + ctx.match_gather_time() + ctx.match_resolve_time(), ctx.verify_kernel_time(), result_bytes

    whole = None if a.skip_whole else timed(pairs)
    out = {"workload": "%d images x %d feats, %s, %d pairs, outlier_frac %.2f" % (a.images, a.feats, a.pairs, len(pairs), a.outlier_frac),
           "shards": a.shards, "scene_generation_s": t_gen, "cuts": {}}
    if whole:
        out["whole_list_ms"], out["whole_list_match_ms"], out["whole_list_verify_ms"] = whole[0], whole[1], whole[2]
        print("whole list: %.1f ms/step (matching %.1f, verification %.1f)" % whole[:3], flush=True)
    for name, lists in cuts.items():
        rows = []
        for r, pl in enumerate(lists):
            wall, km, kv, nbytes = timed(pl)
            rows.append({"shard": r, "pairs": int(len(pl)), "ms": wall, "match_ms": km, "verify_ms": kv, "result_bytes": nbytes})
            print("%s shard %d/%d: %7d pairs  %8.1f ms/step  (matching %.1f, verification %.1f)" % (name, r + 1, a.shards, len(pl), wall, km, kv), flush=True)
        ms = np.array([x["ms"] for x in rows])
        total_bytes = sum(x["result_bytes"] for x in rows)
        gather_ms = 1e3 * total_bytes * (a.shards - 1) / a.shards / (153e9 * 0.6)
        c = {"per_shard": rows, "max_ms": float(ms.max()), "mean_ms": float(ms.mean()), "imbalance_max_over_mean": float(ms.max() / ms.mean()),
             "max_verify_ms": float(max(x["verify_ms"] for x in rows)), "all_gather_estimate_ms": gather_ms,
             "all_gather_note": "ESTIMATE, not measured: %.0f MB of results x (S-1)/S over one 153 GB/s xGMI link direction at 60 %% efficiency" % (total_bytes / 1e6)}
        if whole:
            c["projected_scaling"] = whole[0] / (float(ms.max()) + gather_ms)
            c["projected_scaling_note"] = "T_whole / (max shard + all-gather estimate), all on ONE GPU: unmeasured on 8 GPUs"
            print("%s cut: projected %d-GPU strong scaling %.2fx (max shard %.1f ms, mean %.1f, max verification %.1f, gather est. %.1f ms)" %
                  (name, a.shards, c["projected_scaling"], ms.max(), ms.mean(), c["max_verify_ms"], gather_ms), flush=True)
        out["cuts"][name] = c
    print(json.dumps(out))


if __name__ == "__main__":
    main()
