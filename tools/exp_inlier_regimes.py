"""Verification time against the inlier ratio of the putative matches (the speculation batches
were sized on the synthetic 64 % regime).  outlier_frac f per image => a match is geometrically right with (1 - f)^2."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for f in (0.2, 0.35, 0.5, 0.6):
    scene = synthetic.Scene(n_img, 4096, seed=0, outlier_frac=f)
    ims = [scene.image(i) for i in range(n_img)]
    pairs = synthetic.exhaustive_pairs(n_img)
    ctx = capi.Context(0)
    cams = [capi.simple_pinhole(800., 500., 375., 1000, 750, 1) for _ in range(n_img)]
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    ctx.match_pairs(pairs)
    o = capi.default_two_view_options()
    ctx.verify_pairs(o)
    t0 = time.perf_counter()
    ctx.verify_pairs(o)
    wall = (time.perf_counter() - t0) * 1e3
    tv = ctx.two_view_geometries()
    tr = np.array([list(t.num_trials)[:3] for t in tv])
    ninl = np.array([t.num_inliers for t in tv])
    offs, _ = ctx.matches()
    nm = np.diff(np.array(offs))
    print("outlier_frac %.2f (match inlier ratio %.2f): %d pairs, %.1f matches/pair, median inliers %d, median trials E/F/H %d/%d/%d, "
          "max %d/%d/%d; verify %.1f ms (%.2f us/pair; wall %.1f)" % (f, (1 - f) ** 2, len(pairs), nm.mean(), np.median(ninl), *np.median(tr, axis=0), *tr.max(axis=0),
          ctx.verify_kernel_time(), 1e3 * ctx.verify_kernel_time() / len(pairs), wall), flush=True)
    del ctx
