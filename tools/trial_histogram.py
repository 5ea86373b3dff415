#!/usr/bin/env python3
"""How many trials do the three LO-RANSACs of a pair really run?  Config 2's scene (or --images / --outlier-frac): percentiles of
num_trials / num_models per family and the share of pairs at the family's cap -- what the speculation batch sizes (vp_batch) and the
first-round sizes are chosen against."""
import argparse
import os
import sys
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402

_S = None


def _init(args):
    global _S
    _S = synthetic.Scene(args[0], args[1], seed=0, outlier_frac=args[2])


def _im(i):
    return _S.image(i)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--outlier-frac", type=float, default=0.2)
    a = ap.parse_args()
    args = (a.images, a.feats, a.outlier_frac)
    _init(args)
    with Pool(48, initializer=_init, initargs=(args,)) as pool:
        ims = pool.map(_im, range(a.images), chunksize=4)
    pairs = synthetic.exhaustive_pairs(a.images)
    cams = [capi.simple_pinhole(_S.focal, _S.width / 2.0, _S.height / 2.0, _S.width, _S.height, 1) for _ in range(a.images)]
    ctx = capi.Context(0)
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    ctx.match_pairs(pairs)
    ctx.verify_pairs(capi.default_two_view_options(), user_seed=0)
    tv = ctx.two_view_geometries()
    tr = np.array([list(t.num_trials) for t in tv])
    mo = np.array([list(t.num_models) for t in tv])
    nm = np.array([t.num_matches for t in tv])
    ni = np.array([t.num_inliers for t in tv])
    print("%d pairs; matches per pair: median %d, 90%% %d, max %d; inliers: median %d" % (len(tv), np.median(nm), np.percentile(nm, 90), nm.max(), np.median(ni)))
    caps = [7071, 10000, 1765]
    for f, name in enumerate(("E", "F", "H")):
        t = tr[:, f]
        t = t[t > 0]
        if not len(t):
            continue
        print("%s: trials percentiles 5/25/50/75/95/99 = %s  mean %.1f  at the cap (%d): %.3f of the pairs;  models per trial %.2f" % (
            name, np.percentile(t, [5, 25, 50, 75, 95, 99]).astype(int).tolist(), t.mean(), caps[f], np.mean(t >= caps[f]), mo[:, f].sum() / max(1, t.sum())))
        hist, edges = np.histogram(t, bins=[0, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 1765, 4096, 10001])
        print("   histogram " + "  ".join("%d-%d: %.3f" % (edges[i], edges[i + 1], hist[i] / len(t)) for i in range(len(hist)) if hist[i]))


if __name__ == "__main__":
    main()
