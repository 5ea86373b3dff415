#!/usr/bin/env python3
"""How much does the evaluation order of the 5-point polynomial code matter?  

Runs the same workload through two builds of the library -- ab/lib_generic5pt.so (round 2: constraint matrix and
determinant polynomial by generic polynomial arithmetic) and the current build (the reference's own order of sums
and products, fivept_terms.tbl) -- and counts the pairs whose result differs.  Same seeds, same matches: everything
that differs is caused by the last-place rounding of the E models.

    python tools/compare_fivept_orders.py [--images 500] [--feats 4096] [--outlier-frac 0.2]"""
import argparse
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(a):
    from dagsfm_amd import capi, synthetic
    scene = synthetic.Scene(a.images, a.feats, seed=0, outlier_frac=a.outlier_frac)
    ims = [scene.image(i) for i in range(a.images)]
    pairs = synthetic.exhaustive_pairs(a.images)
    ctx = capi.Context(0)
    cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, 1) for _ in range(a.images)]
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    ctx.match_pairs(pairs)
    ctx.verify_pairs(capi.default_two_view_options(), user_seed=0, stage_filter=False)
    tv = ctx.two_view_geometries()
    ioffs, inl = ctx.inlier_matches()
    np.savez(a.worker, config=np.array([t.config for t in tv]), ninl=np.array([t.num_inliers for t in tv]),
             E=np.array([list(t.E) for t in tv]), F=np.array([list(t.F) for t in tv]), H=np.array([list(t.H) for t in tv]),
             trials=np.array([list(t.num_trials) for t in tv]), models=np.array([list(t.num_models) for t in tv]),
             qvec=np.array([list(t.qvec) for t in tv]), ioffs=np.array(ioffs), inl=np.array(inl), ms=ctx.verify_kernel_time())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--outlier-frac", type=float, default=0.2)
    ap.add_argument("--old", default=os.path.join(ROOT, "ab", "lib_generic5pt.so"))
    ap.add_argument("--worker", default=None)
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    res = []
    with tempfile.TemporaryDirectory() as d:
        for name, lib in (("generic", a.old), ("reference-order", None)):
            env = dict(os.environ)
            if lib:
                env["DSM_LIB_PATH"] = lib
            else:
                env.pop("DSM_LIB_PATH", None)
            out = os.path.join(d, name + ".npz")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--images", str(a.images), "--feats", str(a.feats),
                                   "--outlier-frac", str(a.outlier_frac), "--worker", out], env=env)
            res.append(dict(np.load(out)))
    g, r = res
    n = len(g["config"])
    same_inl = np.zeros(n, bool)
    for k in range(n):
        x = g["inl"][g["ioffs"][k]:g["ioffs"][k + 1]]
        y = r["inl"][r["ioffs"][k]:r["ioffs"][k + 1]]
        same_inl[k] = x.shape == y.shape and (x == y).all()
    calibrated = (g["trials"][:, 0] > 0)
    print("workload: %d images x %d features, outlier_frac %.2f (match inlier ratio %.2f), %d pairs, calibrated E + F + H + pose" %
          (a.images, a.feats, a.outlier_frac, (1 - a.outlier_frac) ** 2, n))
    print("verification ms: generic order %.1f, reference order %.1f" % (float(g["ms"]), float(r["ms"])))
    print("pairs that ran the E family:                        %d" % calibrated.sum())
    print("pairs whose E matrix differs in any bit:            %d" % (g["E"] != r["E"]).any(axis=1).sum())
    print("   largest |dE| (E has unit Frobenius norm):        %.3g" % np.abs(g["E"] - r["E"]).max())
    print("pairs whose E trial count differs:                  %d" % (g["trials"][:, 0] != r["trials"][:, 0]).sum())
    print("pairs whose E model count differs:                  %d" % (g["models"][:, 0] != r["models"][:, 0]).sum())
    print("pairs whose F or H differs in any bit:              %d" % ((g["F"] != r["F"]).any(axis=1) | (g["H"] != r["H"]).any(axis=1)).sum())
    print("pairs whose configuration differs:                  %d" % (g["config"] != r["config"]).sum())
    print("pairs whose inlier count differs:                   %d" % (g["ninl"] != r["ninl"]).sum())
    print("pairs whose inlier match list differs:              %d" % (~same_inl).sum())
    print("pairs whose qvec differs by > 1e-6:                 %d" % (np.abs(g["qvec"] - r["qvec"]).max(axis=1) > 1e-6).sum())


if __name__ == "__main__":
    main()
