#!/usr/bin/env python3
"""Differential fuzz of the HIP descriptor matcher (dsm_match_pairs: K1 tile maxima, K1b resolve, gathered second pass,
cross check) against the CPU oracle's MatchSiftFeaturesCPU restatement: images of awkward sizes (0, 1, tile edges
31..33, 63..65, 127..129, a few hundred rows), descriptors with exact duplicates inside an image and across images,
near-duplicates, all-zero and saturated rows, under random ratio / distance / cross-check options.  Every pair's match
list must be identical.

  python tools/fuzz_match.py [--batches 6] [--images 60] [--pairs 1500] [--seed 1] [--workers 64]

Test infrastructure: the oracle is the checker here, as in tests/."""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi  # noqa: E402

SIZES = [0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300, 500, 700]


def sift_like(rng, n):
    """Non-negative vectors of L2 norm ~512 rounded to u8, like the reference's descriptors (sift.cc:1109-1133)."""
    v = rng.gamma(0.6, 1.0, size=(n, 128))
    v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-9)
    return np.clip(np.rint(v * 512.0), 0, 255).astype(np.uint8)


def make_images(seed, n_images):
    rng = np.random.default_rng(seed)
    pool = sift_like(rng, 400)
    imgs = []
    for _ in range(n_images):
        n = int(rng.choice(SIZES))
        d = np.zeros((n, 128), np.uint8)
        if n:
            src = rng.random(n)
            k = rng.integers(0, len(pool), n)
            noisy = pool[k].astype(np.int32) + rng.integers(-12, 13, (n, 128)) * (rng.random((n, 1)) < 0.7)
            d = np.where((src < 0.6)[:, None], np.clip(noisy, 0, 255), sift_like(rng, n)).astype(np.uint8)
            exact = src > 0.92          # exact copies of pool entries: equal dot products in many images
            d[exact] = pool[k[exact]]
            if n > 3 and rng.random() < 0.5:   # duplicates inside the image: best == second best for their partners
                a, b = rng.integers(0, n, 2)
                d[a] = d[b]
            if rng.random() < 0.15:
                d[rng.integers(n)] = 0
            if rng.random() < 0.1:
                d[rng.integers(n)] = 255
        imgs.append(d)
    return imgs


_O = None


def _init():
    global _O
    from tests import oracle_lib
    _O = oracle_lib.load()


def _oracle_one(a):
    d1, d2, ratio, dist, cross = a
    return _O.match_sift_features_cpu(d1, d2, ratio, dist, cross)


def run_fuzz(ctx, batches, n_images, n_pairs, seed, workers, log=print):
    bad = total = n_matches = 0
    with Pool(workers, initializer=_init) as pool:
        for b in range(batches):
            rng = np.random.default_rng([seed, b])
            ratio = float(rng.choice([0.6, 0.8, 0.95, 1.0]))
            dist = float(rng.choice([0.4, 0.7, 1.0, 1.6]))
            cross = bool(rng.random() < 0.7)
            imgs = make_images(int(rng.integers(0, 2**31)), n_images)
            i = rng.integers(0, n_images, n_pairs)
            j = (i + rng.integers(1, n_images, n_pairs)) % n_images
            pairs = np.stack([i, j], axis=1).astype(np.uint32)
            t0 = time.perf_counter()
            ctx.set_images(imgs)
            opts = capi.default_match_options(max_ratio=ratio, max_distance=dist, cross_check=int(cross))
            ctx.match_pairs(pairs, opts)
            offs, m = ctx.matches()
            t_dev = time.perf_counter() - t0
            t0 = time.perf_counter()
            refs = pool.map(_oracle_one, [(imgs[int(a)], imgs[int(c)], ratio, dist, cross) for a, c in pairs], chunksize=8)
            t_or = time.perf_counter() - t0
            nb = 0
            for k, r in enumerate(refs):
                g = m[int(offs[k]):int(offs[k + 1])]
                n_matches += len(r)
                if g.shape != r.shape or not (g == r).all():
                    nb += 1
                    if nb <= 10:
                        log("MISMATCH batch %d pair %d (images %d x %d rows): device %d matches, oracle %d" %
                            (b, k, len(imgs[int(pairs[k][0])]), len(imgs[int(pairs[k][1])]), len(g), len(r)))
            bad += nb
            total += len(pairs)
            log("batch %d: %d pairs, ratio %.2f distance %.1f cross_check %d: %d mismatches (device %.2f s, oracle %.1f s)" %
                (b, len(pairs), ratio, dist, cross, nb, t_dev, t_or))
    return total, bad, n_matches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--images", type=int, default=60)
    ap.add_argument("--pairs", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--workers", type=int, default=min(96, os.cpu_count() or 8))
    args = ap.parse_args()
    ctx = capi.Context(0)
    total, bad, nm = run_fuzz(ctx, args.batches, args.images, args.pairs, args.seed, args.workers, log=lambda s: print(s, flush=True))
    print("FUZZ RESULT: %d pairs (%d matches), %d mismatches" % (total, nm, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
