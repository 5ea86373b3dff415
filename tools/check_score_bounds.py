#!/usr/bin/env python3
"""The bound step of the scoring (k_prescore) checked on EVERY slot of a whole workload: with
DSM_SCORE_PREFILTER=check every (model, pair) slot is scored exactly and its exact inlier count must lie inside the bound
step's [lower, upper]; prints the violations (must be 0) and how many slots the filter skips.
    python tools/check_score_bounds.py [--images 500] [--outlier-frac 0.2] [--uncalibrated]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DSM_SCORE_PREFILTER"] = "check"
from dagsfm_amd import capi, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=500)
ap.add_argument("--feats", type=int, default=4096)
ap.add_argument("--outlier-frac", type=float, default=0.2)
ap.add_argument("--uncalibrated", action="store_true")
a = ap.parse_args()
scene = synthetic.Scene(a.images, a.feats, seed=0, outlier_frac=a.outlier_frac)
ims = [scene.image(i) for i in range(a.images)]
pairs = synthetic.exhaustive_pairs(a.images)
cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, 0 if a.uncalibrated else 1) for _ in range(a.images)]
ctx = capi.Context(0)
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
ctx.match_pairs(pairs)
ctx.verify_pairs(capi.default_two_view_options(), user_seed=0, stage_filter=True)
c = ctx.debug_verify_counters()
tv = ctx.two_view_geometries()
models = np.array([list(t.num_models) for t in tv], dtype=np.int64).sum(axis=0)  # E, F, H, T: minimal-sample + LO models scored
# (the reports count the models up to each pair's stopping trial; the scoring kernels also see the speculated trials behind it,
# so the skipped slots can outnumber the reported models)
print("%d images, outlier_frac %.2f, %s: %d pairs, models in the reports E / F / H %d / %d / %d; bound violations %d; slots the bound step skips %d" % (
    a.images, a.outlier_frac, "uncalibrated" if a.uncalibrated else "calibrated", len(pairs), models[0], models[1], models[2], int(c[14]), int(c[15])))
sys.exit(1 if c[14] else 0)
