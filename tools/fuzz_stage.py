#!/usr/bin/env python3
"""End-to-end differential fuzz of the stage calls on ONE context: many small seeded scenes (general / planar /
panoramic; calibrated or not; several camera models), each through
    dsm_set_images -> dsm_match_pairs -> dsm_verify_pairs -> dsm_guided_match_pairs (+ Match()'s post-filter)
with random matching / verification options, compared pair by pair with the oracle: match lists against
MatchSiftFeaturesCPU, two-view records against TwoViewGeometry::Estimate on those matches, the guided inlier lists
against MatchGuidedSiftFeaturesCPU (sift.cc:824-875) and the post-filter of matching.cc:441-470 / 828-831.

  python tools/fuzz_stage.py [--scenes 40] [--seed 1] [--workers 64]

Test infrastructure: the oracle is the checker here, as in tests/."""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_verify import CAMS, poison_device_memory, rec_diff  # noqa: E402

_O = None


def _init():
    global _O
    from tests import oracle_lib
    _O = oracle_lib.load()


def _cam(spec):
    return capi.camera(spec[0], spec[1], 1000, 750, spec[2])


def _oracle_pair(a):
    """Everything the oracle says about one pair: matches, record, inliers, guided list (or None)."""
    d1, k1, c1, d2, k2, c2, mkw, okw, seed, guided = a
    m = _O.match_sift_features_cpu(d1, d2, mkw["max_ratio"], mkw["max_distance"], bool(mkw["cross_check"]))
    opts = capi.default_two_view_options(**okw)
    ref, inl = _O.estimate_two_view_geometry(_cam(c1), k1.astype(np.float64), _cam(c2), k2.astype(np.float64), m, opts, seed)
    g = None
    if guided and ref.num_inliers >= opts.min_num_inliers:
        g = _O.match_guided_sift_features_cpu(k1, k2, d1, d2, ref, max_error=opts.max_error, max_ratio=mkw["max_ratio"],
                                              max_distance=mkw["max_distance"], cross_check=bool(mkw["cross_check"]))
    return m, bytes(ref), inl, g


def run_fuzz(ctx, n_scenes, seed, workers, log=print, poison=False):
    bad = total = 0
    stats = dict(guided=0, kept=0)
    with Pool(workers, initializer=_init) as pool:
        for s in range(n_scenes):
            rng = np.random.default_rng([seed, s])
            n_img = int(rng.integers(3, 8))
            feats = int(rng.choice([40, 150, 300, 500]))
            kind = str(rng.choice(["general", "general", "planar", "panoramic"]))
            camspec = CAMS[int(rng.integers(len(CAMS)))] if rng.random() < 0.4 else CAMS[0]
            prior = bool(rng.random() < 0.6)
            scene = synthetic.Scene(n_img, feats, seed=int(rng.integers(0, 2**31)), n_pool=int(feats * rng.uniform(1.05, 2.0)),
                                    planar=(kind == "planar"), panoramic=(kind == "panoramic"), outlier_frac=float(rng.choice([0.1, 0.2, 0.5])),
                                    camera=(camspec[0], camspec[1]))
            ims = [scene.image(i) for i in range(n_img)]
            cs = (camspec[0], camspec[1], prior)
            pairs = synthetic.exhaustive_pairs(n_img)
            if rng.random() < 0.5:
                pairs = pairs[rng.permutation(len(pairs))[:max(1, len(pairs) // 2)]]
            mkw = dict(max_ratio=float(rng.choice([0.7, 0.8, 0.9])), max_distance=float(rng.choice([0.6, 0.7, 1.0])),
                       cross_check=int(rng.random() < 0.7))
            okw = dict(max_error=float(rng.choice([2.0, 4.0, 8.0])), confidence=float(rng.choice([0.99, 0.999])),
                       max_num_trials=int(rng.choice([500, 10000])), min_num_inliers=int(rng.choice([8, 15, 25])),
                       min_inlier_ratio=float(rng.choice([0.1, 0.25])), detect_watermark=int(rng.random() < 0.8))
            guided = bool(rng.random() < 0.6)
            user_seed = int(rng.integers(0, 1000))
            if poison and s % 4 == 0:
                poison_device_memory(256)
            t0 = time.perf_counter()
            ctx.set_images([im[0] for im in ims], [im[1] for im in ims], [_cam(cs)] * n_img)
            mo = capi.default_match_options(**mkw)
            opts = capi.default_two_view_options(**okw)
            ctx.match_pairs(pairs, mo)
            offs, m = ctx.matches()
            ctx.verify_pairs(opts, user_seed=user_seed, stage_filter=not guided)
            if guided:
                ctx.guided_match_pairs(mo, opts, stage_filter=True)
            tvgs = ctx.two_view_geometries()
            ioffs, im_ = ctx.inlier_matches()
            t_dev = time.perf_counter() - t0
            refs = pool.map(_oracle_pair, [(ims[int(i)][0], ims[int(i)][1], cs, ims[int(j)][0], ims[int(j)][1], cs, mkw, okw,
                                            capi.pair_seed(int(i), int(j), user_seed), guided) for i, j in pairs], chunksize=1)
            nb = 0
            for k, (rm, rb, rinl, rg) in enumerate(refs):
                r = capi.TwoViewGeometry.from_buffer_copy(rb)
                got, gm = tvgs[k], m[int(offs[k]):int(offs[k + 1])]
                ginl = im_[int(ioffs[k]):int(ioffs[k + 1])]
                d = None
                if gm.shape != rm.shape or not (gm == rm).all():
                    d = "matches (%d vs %d)" % (len(gm), len(rm))
                else:
                    exp = rg if rg is not None else rinl
                    stats["guided"] += rg is not None
                    if len(exp) < opts.min_num_inliers:  # Match()'s post-filter (matching.cc:828-831): a default record
                        if not (got.config == 0 and got.num_inliers == 0 and len(ginl) == 0):
                            d = "filtered pair not empty (config %d, %d inliers)" % (got.config, got.num_inliers)
                    else:
                        stats["kept"] += 1
                        if rg is None:
                            d = rec_diff(got, r)
                        elif got.config != r.config or got.num_inliers != len(exp):
                            d = "guided: config %d vs %d, inliers %d vs %d" % (got.config, r.config, got.num_inliers, len(exp))
                        else:
                            for name in ("E", "F", "H"):
                                if not (np.array(getattr(got, name)) == np.array(getattr(r, name))).all():
                                    d = "guided: %s differs" % name
                        if d is None and (ginl.shape != exp.shape or not (ginl == exp).all()):
                            d = "inlier matches"
                if d is not None:
                    nb += 1
                    if nb <= 5:
                        log("MISMATCH scene %d pair %d (%d,%d): %s" % (s, k, int(pairs[k][0]), int(pairs[k][1]), d))
            bad += nb
            total += len(pairs)
            log("scene %d: %s, %d images x %d feats, camera %d prior %d, %d pairs, guided %d, %s %s: %d mismatches (device %.2f s)" %
                (s, kind, n_img, feats, camspec[0], prior, len(pairs), guided, mkw, okw, nb, t_dev))
    return total, bad, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--workers", type=int, default=min(64, os.cpu_count() or 8))
    ap.add_argument("--poison", action="store_true", help="fill freed device memory with 0xFF every fourth scene")
    args = ap.parse_args()
    ctx = capi.Context(0)
    total, bad, stats = run_fuzz(ctx, args.scenes, args.seed, args.workers, log=lambda s: print(s, flush=True), poison=args.poison)
    print("FUZZ RESULT: %d pairs (%d kept, %d with a guided list), %d mismatches" % (total, stats["kept"], stats["guided"], bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
