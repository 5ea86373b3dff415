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
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs in the CPU-baseline sample (-1 = auto, 0 = skip)")
    return ap.parse_args()


def cpu_baseline(scene_images, pairs, n_sample, verify):
    """Times the CPU oracle (the reference algorithm restated, oracle/) on a bounded sample of
    the same workload, using all host cores like the reference's matcher/verifier thread pools
    (/root/reference/src/feature/matching.cc:640-674)."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import oracle_lib
    orc = oracle_lib.load()
    cores = os.cpu_count() or 1
    sample = pairs[np.linspace(0, len(pairs) - 1, n_sample).astype(np.int64)]

    def one(p):
        d1, d2 = scene_images[int(p[0])][0], scene_images[int(p[1])][0]
        m = orc.match_sift_features_cpu(d1, d2)
        return len(m)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(one, sample))
    dt = time.perf_counter() - t0
    return {"value": len(sample) / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d of %d pairs (%s), %d threads, oracle/ restatement of the reference CPU path, %.1f s"
                      % (len(sample), len(pairs), "match only" if not verify else "match + verify", cores, dt)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from dagsfm_amd import capi, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    verify = False  # two-view verification is wired in below once available
    scene = synthetic.Scene(args.images, args.feats, seed=args.seed)
    images = [scene.image(i) for i in range(args.images)]
    pairs = synthetic.exhaustive_pairs(args.images)
    # strong scaling: contiguous block of the pair list per rank
    bounds = np.linspace(0, len(pairs), world + 1).astype(np.int64)
    my_pairs = pairs[bounds[rank]:bounds[rank + 1]]

    ctx = capi.Context(local_rank)
    ctx.set_images([im[0] for im in images], [im[1] for im in images])
    opts = capi.default_match_options()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_results():
        """All-gather of the per-pair match graph (counts + matches) over RCCL."""
        counts = torch.empty(len(my_pairs), dtype=torch.int32, device=dev)
        capi.lib().dsm_get_match_counts(ctx._h, counts.data_ptr())
        offs = torch.empty(len(my_pairs) + 1, dtype=torch.int64, device=dev)
        capi.lib().dsm_get_matches(ctx._h, offs.data_ptr(), None, 0)
        total = int(offs[-1].item())
        if world == 1:
            return int(counts.sum().item()), total
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([total], dtype=torch.int64, device=dev))
        mx = int(sizes.max().item())
        mine = torch.zeros((mx, 2), dtype=torch.int32, device=dev)
        if total:
            capi.lib().dsm_get_matches(ctx._h, None, mine.data_ptr(), total)
        allm = torch.empty((world * mx, 2), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allm, mine)
        maxp = int(np.diff(bounds).max())
        cpad = torch.zeros(maxp, dtype=torch.int32, device=dev)
        cpad[:len(my_pairs)] = counts
        allc = torch.empty(world * maxp, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allc, cpad)
        return int(allc.sum().item()), int(sizes.sum().item())

    def step():
        ctx.match_pairs(my_pairs, opts)
        return gather_results()

    for _ in range(args.warmup):
        step()
    k1_ms, k1_launches = 0.0, 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_matched, n_matches = step()
        ms, nl = ctx.match_kernel_time()
        k1_ms += ms
        k1_launches += nl
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
        out = {
            "metric": "verified image-pairs/sec at 4096 feats/image" if verify else
                      "matched image-pairs/sec at %d feats/image (verification not in this build)" % args.feats,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8 (int8 MFMA, int32 accumulate)" if not verify else "u8+f64", "data": "synthetic",
            "config": {"workload": "%d images x %d feats exhaustive (%d pairs), %s" % (
                args.images, args.feats, n_pairs, "match + two-view RANSAC" if verify else "match only"),
                "pairs": n_pairs, "total_matches": n_matches, "parallelism": "pair-sharded x%d + RCCL all-gather" % world},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": INT8_MFMA_DENSE_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": achieved / INT8_MFMA_DENSE_PEAK, "traffic": None,
                         "kernel": "k1_best_rows", "avg_launch_ms": 1e3 * avg_launch_s, "launches": k1_launches,
                         "note": "int8 ops (2 per MAC) counted as flops; algorithmic 2*128*N^2 per pair"},
        }
        n_cpu = args.cpu_pairs
        if n_cpu < 0:
            n_cpu = 24 if args.feats >= 2048 else 200
        if world == 1 and n_cpu > 0:
            out["cpu_baseline"] = cpu_baseline(images, pairs, min(n_cpu, len(pairs)), verify)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
