#!/usr/bin/env python3
"""Differential fuzz of the HIP two-view verification against the CPU oracle: thousands of small seeded pairs of very
different structure (general / planar / pure rotation / watermark / collinear / repeated points / pure outliers / almost
no matches; all camera models; calibrated and not), under random option sets, through the stage calls
(dsm_set_images + dsm_set_matches + dsm_verify_pairs), every record and inlier list compared with the oracle's
(tests/test_verify_gpu.py::tvg_equal).  A mismatch prints the batch seed and the pair so that it can be replayed.

  python tools/fuzz_verify.py [--batches 6] [--pairs 1500] [--seed 1] [--workers 64]

Test infrastructure: the oracle is the checker here, as in tests/."""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402

CAMS = [
    (0, [800.0, 500.0, 375.0]),
    (1, [800.0, 810.0, 500.0, 375.0]),
    (2, [800.0, 500.0, 375.0, 0.04]),
    (3, [800.0, 500.0, 375.0, 0.05, 0.01]),
    (4, [800.0, 805.0, 500.0, 375.0, -0.12, 0.05, -0.001, 0.001]),
    (5, [800.0, 805.0, 500.0, 375.0, -0.05, 0.01, -0.001, 0.001]),
    (6, [800.0, 805.0, 500.0, 375.0, -0.12, 0.05, -0.001, 0.001, 0.001, 0.02, -0.02, 0.001]),
    (7, [800.0, 805.0, 500.0, 375.0, 0.4]),
    (8, [800.0, 500.0, 375.0, 0.03]),
    (9, [800.0, 500.0, 375.0, 0.03, 0.005]),
    (10, [800.0, 805.0, 500.0, 375.0, -0.05, 0.01, -0.001, 0.001, 0.001, 0.002, -0.002, 0.001]),
]
W, H = 1000, 750
KINDS = ("general", "general", "general", "planar", "rotation", "watermark", "collinear", "repeated", "outliers", "tiny", "two_motions")


def rot(rng, mag):
    a = rng.normal(size=3)
    a = a / np.linalg.norm(a) * rng.uniform(0, mag)
    th = np.linalg.norm(a)
    if th < 1e-12:
        return np.eye(3)
    k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


BIG = False  # --big: hundreds to thousands of matches per pair (many rounds, batch growth, points beyond the LDS staging)


def make_pair(seed):
    """One problem: (cam1, cam2, kp1 float32 [n1,2], kp2, matches uint32 [m,2], kind)."""
    rng = np.random.default_rng(seed)
    kind = KINDS[rng.integers(len(KINDS))]
    c1 = CAMS[rng.integers(len(CAMS))] if rng.random() < 0.5 else CAMS[0]
    c2 = CAMS[rng.integers(len(CAMS))] if rng.random() < 0.3 else c1
    prior1, prior2 = (rng.random() < 0.6), (rng.random() < 0.6)
    if rng.random() < 0.7:
        prior2 = prior1
    n = int(rng.choice([0, 3, 6, 8, 14, 15, 16, 25, 40, 80, 150, 300], p=[.02, .02, .03, .03, .05, .05, .05, .15, .2, .2, .15, .05]))
    if BIG:
        n = int(rng.choice([300, 600, 1200, 1600, 2500]))
    if kind == "tiny":
        n = int(rng.integers(0, 12))
    noise = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
    out_frac = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.7, 0.85]))
    if kind in ("general", "planar", "rotation", "tiny", "two_motions"):
        X = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(-1.1, 1.1, n), rng.uniform(3.0, 7.0, n)]
        if kind == "planar":
            nrm = np.array([rng.normal(scale=0.2), rng.normal(scale=0.2), 1.0])
            X[:, 2] = (5.0 - X[:, 0] * nrm[0] - X[:, 1] * nrm[1]) / nrm[2]
        R = rot(rng, 0.35)
        t = np.zeros(3) if kind == "rotation" else rng.normal(size=3) * rng.uniform(0.05, 0.8)
        Y = X @ R.T + t
        if kind == "two_motions":  # a second rigid motion for part of the points (EstimateMultiple finds both)
            g = rng.random(n) < rng.uniform(0.3, 0.5)
            R2, t2 = rot(rng, 0.35), rng.normal(size=3) * 0.6
            Y[g] = X[g] @ R2.T + t2
        ok = Y[:, 2] > 0.5
        X, Y = X[ok], Y[ok]
        n = len(X)
        x1, y1 = synthetic.world_to_image(c1[0], c1[1], X[:, 0] / X[:, 2], X[:, 1] / X[:, 2])
        x2, y2 = synthetic.world_to_image(c2[0], c2[1], Y[:, 0] / Y[:, 2], Y[:, 1] / Y[:, 2])
        p1 = np.c_[x1, y1] + rng.normal(scale=noise, size=(n, 2))
        p2 = np.c_[x2, y2] + rng.normal(scale=noise, size=(n, 2))
        k = int(out_frac * n)
        if k:
            sel = rng.choice(n, k, replace=False)
            p2[sel] = np.c_[rng.uniform(0, W, k), rng.uniform(0, H, k)]
    elif kind == "watermark":
        p1 = np.c_[rng.uniform(2, 70, n), rng.uniform(5, H - 5, n)]
        p2 = p1 + rng.uniform(-6, 6, 2) + rng.normal(scale=min(noise, 0.4), size=(n, 2))
        k = int(min(out_frac, 0.3) * n)
        if k:
            sel = rng.choice(n, k, replace=False)
            p2[sel] = np.c_[rng.uniform(0, W, k), rng.uniform(0, H, k)]
    elif kind == "collinear":
        s = rng.uniform(0, 1, n)
        a, b = rng.uniform(0, W, 2), rng.uniform(0, H, 2)
        p1 = np.c_[a[0] + s * (a[1] - a[0]), b[0] + s * (b[1] - b[0])]
        p2 = p1 * rng.uniform(0.8, 1.2) + rng.uniform(-20, 20, 2) + rng.normal(scale=noise * 0.2, size=(n, 2))
    elif kind == "repeated":
        d = max(1, int(rng.integers(1, 7)))
        base1 = np.c_[rng.uniform(0, W, d), rng.uniform(0, H, d)]
        base2 = base1 + rng.uniform(-30, 30, 2)
        idx = rng.integers(0, d, n)
        p1, p2 = base1[idx], base2[idx]
    else:  # outliers
        p1 = np.c_[rng.uniform(0, W, n), rng.uniform(0, H, n)]
        p2 = np.c_[rng.uniform(0, W, n), rng.uniform(0, H, n)]
    n = len(p1)
    # keypoint lists: the matched points in shuffled positions plus unmatched extras; a few matches name a keypoint twice
    e1, e2 = int(rng.integers(0, 6)), int(rng.integers(0, 6))
    kp1 = np.r_[p1, np.c_[rng.uniform(0, W, e1), rng.uniform(0, H, e1)]].astype(np.float32)
    kp2 = np.r_[p2, np.c_[rng.uniform(0, W, e2), rng.uniform(0, H, e2)]].astype(np.float32)
    perm1, perm2 = rng.permutation(len(kp1)), rng.permutation(len(kp2))
    inv1, inv2 = np.argsort(perm1), np.argsort(perm2)
    kp1, kp2 = kp1[perm1], kp2[perm2]
    m = np.stack([inv1[:n], inv2[:n]], axis=1).astype(np.uint32)
    if n > 4 and rng.random() < 0.15:
        m[rng.integers(n)] = m[rng.integers(n)]          # an exact duplicate of a match
        m[rng.integers(n), 1] = m[rng.integers(n), 1]    # two matches into one keypoint
    if len(kp1) == 0:
        kp1 = np.zeros((1, 2), np.float32)
    if len(kp2) == 0:
        kp2 = np.zeros((1, 2), np.float32)
    return (c1[0], c1[1], prior1), (c2[0], c2[1], prior2), kp1, kp2, m, kind


def make_options(rng, rng2):
    kw = dict(max_error=float(rng.choice([0.1, 0.7, 2.0, 4.0, 9.0, 40.0])),  # 0.1: below the H bound step's range (T < 2^-6)
              confidence=float(rng.choice([0.9, 0.99, 0.999, 0.9999])),
              max_num_trials=int(rng.choice([50, 400, 2000, 10000])),
              min_inlier_ratio=float(rng.choice([0.1, 0.25, 0.5])),
              min_num_inliers=int(rng.choice([0, 8, 15, 30])),
              min_num_trials=int(rng.choice([0, 0, 30, 200])),
              detect_watermark=int(rng.random() < 0.8),
              min_E_F_inlier_ratio=float(rng.choice([0.9, 0.95, 0.99])),
              max_H_inlier_ratio=float(rng.choice([0.6, 0.8, 0.95])),
              multiple_models=int(rng2.random() < 0.35), multiple_ignore_watermark=int(rng2.random() < 0.5))
    if kw["min_num_trials"] > kw["max_num_trials"]:
        kw["min_num_trials"] = kw["max_num_trials"]
    return kw


_O = None


def _init():
    global _O
    from tests import oracle_lib
    _O = oracle_lib.load()


def _oracle_one(a):
    (m1, q1, pr1), (m2, q2, pr2), kp1, kp2, m, okw, seed = a
    cam1, cam2 = capi.camera(m1, q1, W, H, pr1), capi.camera(m2, q2, W, H, pr2)
    opts = capi.default_two_view_options(**okw)
    ref, inl = _O.estimate_two_view_geometry(cam1, kp1.astype(np.float64), cam2, kp2.astype(np.float64), m, opts, seed)
    return bytes(ref), inl


def rec_diff(g, r):
    if g.config != r.config:
        return "config %d vs %d" % (g.config, r.config)
    if g.num_inliers != r.num_inliers:
        return "num_inliers %d vs %d" % (g.num_inliers, r.num_inliers)
    if list(g.num_trials) != list(r.num_trials):
        return "num_trials %s vs %s" % (list(g.num_trials), list(r.num_trials))
    if list(g.num_models) != list(r.num_models):
        return "num_models %s vs %s" % (list(g.num_models), list(r.num_models))
    for name in ("E", "F", "H"):
        a, b = np.array(getattr(g, name)), np.array(getattr(r, name))
        if not ((a == b) | (np.isnan(a) & np.isnan(b))).all():
            return "%s differs" % name
    if not np.allclose(np.array(g.qvec), np.array(r.qvec), rtol=1e-6, atol=1e-12, equal_nan=True):
        return "qvec"
    if not np.allclose(np.array(g.tvec), np.array(r.tvec), rtol=1e-6, atol=1e-12, equal_nan=True):
        return "tvec"
    if not (abs(g.tri_angle - r.tri_angle) <= 1e-6 * max(abs(r.tri_angle), 1e-9) or (np.isnan(g.tri_angle) and np.isnan(r.tri_angle))):
        return "tri_angle %r vs %r" % (g.tri_angle, r.tri_angle)
    return None


def poison_device_memory(megabytes=768):
    """Leaves 0xFF in device memory the allocator is about to hand out again: a second context uploads all-0xFF descriptors
    and keypoints and is closed.  Buffers the product grows afterwards land (in practice) on these pages, so a kernel that
    reads what it never wrote sees NaNs, huge counters and 'active' flags instead of the zeros of fresh pages."""
    c = capi.Context(0)
    rows = 1 << 16
    n = max(1, megabytes * (1 << 20) // (rows * (128 + 8)))
    d = np.full((rows, 128), 255, np.uint8)
    k = np.full((rows, 2), np.float32(np.nan))
    c.set_images([d] * n, [k] * n, [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True)] * n)
    c.sync()
    c.close()


def run_fuzz(ctx, batches, pairs_per_batch, seed, workers, log=print, first_batch=0, grow=False, poison=False):
    """Returns (pairs checked, mismatches, {config: count}).  grow: batch b has (b + 1) x pairs_per_batch pairs, so every
    call outgrows the buffers of the one before; poison: poison_device_memory() before every call."""
    bad = 0
    total = 0
    configs = {}
    with Pool(workers, initializer=_init) as pool:
        for b in range(first_batch, first_batch + batches):
            rng = np.random.default_rng([seed, b])
            okw = make_options(rng, np.random.default_rng([seed, b, 7]))  # (second stream: fields added later leave the earlier draws alone)
            user_seed = int(rng.integers(0, 1000))
            n_here = pairs_per_batch * (b - first_batch + 1) if grow else pairs_per_batch
            probs = pool.map(make_pair, [int(s) for s in rng.integers(0, 2**31, n_here)], chunksize=16)
            if poison:
                poison_device_memory()
            descs, kps, cams, pairs, matches = [], [], [], [], []
            for k, (c1, c2, kp1, kp2, m, kind) in enumerate(probs):
                for c, kp in ((c1, kp1), (c2, kp2)):
                    descs.append(np.zeros((len(kp), 128), np.uint8))
                    kps.append(kp)
                    cams.append(capi.camera(c[0], c[1], W, H, c[2]))
                pairs.append((2 * k, 2 * k + 1))
                matches.append(m)
            pairs = np.array(pairs, dtype=np.uint32)
            t0 = time.perf_counter()
            ctx.set_images(descs, kps, cams)
            ctx.set_matches(pairs, matches)
            opts = capi.default_two_view_options(**okw)
            ctx.verify_pairs(opts, user_seed=user_seed, stage_filter=False)
            tvgs = ctx.two_view_geometries()
            ioffs, im = ctx.inlier_matches()
            t_dev = time.perf_counter() - t0
            t0 = time.perf_counter()
            refs = pool.map(_oracle_one, [(p[0], p[1], p[2], p[3], p[4], okw, capi.pair_seed(2 * k, 2 * k + 1, user_seed))
                                          for k, p in enumerate(probs)], chunksize=4)
            t_or = time.perf_counter() - t0
            nb = 0
            for k, (rb, rinl) in enumerate(refs):
                r = capi.TwoViewGeometry.from_buffer_copy(rb)
                configs[r.config] = configs.get(r.config, 0) + 1
                d = rec_diff(tvgs[k], r)
                if d is None and not (im[int(ioffs[k]):int(ioffs[k + 1])] == rinl).all():
                    d = "inlier matches"
                if d is not None:
                    nb += 1
                    if nb <= 10:
                        log("MISMATCH batch %d pair %d (%s, %d matches, cams %d/%d prior %d/%d): %s" %
                            (b, k, probs[k][5], len(probs[k][4]), probs[k][0][0], probs[k][1][0], probs[k][0][2], probs[k][1][2], d))
            bad += nb
            total += len(probs)
            log("batch %d: %d pairs, options %s user_seed %d: %d mismatches (device %.2f s, oracle %.1f s)" %
                (b, len(probs), okw, user_seed, nb, t_dev, t_or))
    return total, bad, configs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--pairs", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--workers", type=int, default=min(96, os.cpu_count() or 8))
    ap.add_argument("--big", action="store_true", help="300 .. 2 500 matches per pair")
    ap.add_argument("--grow", action="store_true", help="batch b has (b + 1) x --pairs pairs")
    ap.add_argument("--poison", action="store_true", help="fill freed device memory with 0xFF before every call")
    args = ap.parse_args()
    global BIG
    BIG = args.big  # (set before the pool forks)
    ctx = capi.Context(0)
    total, bad, configs = run_fuzz(ctx, args.batches, args.pairs, args.seed, args.workers, log=lambda m: print(m, flush=True),
                                   grow=args.grow, poison=args.poison)
    print("configurations seen (oracle): %s" % dict(sorted(configs.items())))
    print("FUZZ RESULT: %d pairs, %d mismatches" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
