#!/usr/bin/env python3
"""Writes the pair-level golden vectors under tests/golden/ (the reference has no test that pins
TwoViewGeometry::Estimate / SiftFeatureMatcher::Match, so the pinned CPU oracle -- oracle/, checked against the
reference's own known-answer tests in tests/test_oracle_*.py -- produces them with a fixed per-pair seed).
Inputs AND outputs are stored, so the files do not depend on the synthetic generator:

    python tools/make_golden.py          # rewrites tests/golden/pairs_v2.npz (and pairs_v1.npz with --v1)

tests/test_golden.py checks that the oracle still reproduces them (CPU) and that the HIP path does (GPU).
Regenerate only when the oracle's defined arithmetic changes on purpose, and say so in the commit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import capi, synthetic  # noqa: E402
from tests import oracle_lib  # noqa: E402

TVG_FIELDS = ("config", "num_inliers", "num_matches", "F", "E", "H", "qvec", "tvec", "tri_angle", "num_trials", "num_models")


def tvg_to_arrays(t):
    return {k: np.array(getattr(t, k)) for k in TVG_FIELDS}


def main():
    orc = oracle_lib.load()
    out = {}
    cases = []

    def add(name, d1, d2, k1, k2, prior, seed, multiple=0, min_num_inliers=15):
        cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior)
        opts = capi.default_two_view_options()
        opts.multiple_models = multiple
        opts.min_num_inliers = min_num_inliers
        m = orc.match_sift_features_cpu(d1, d2)
        tv, inl = orc.estimate_two_view_geometry(cam, k1, cam, k2, m, opts, seed)
        out[name + "/desc1"], out[name + "/desc2"] = d1, d2
        out[name + "/kp1"], out[name + "/kp2"] = k1, k2
        out[name + "/params"] = np.array([prior, seed, multiple, min_num_inliers], dtype=np.int64)
        out[name + "/matches"], out[name + "/inlier_matches"] = m, inl
        for k, v in tvg_to_arrays(tv).items():
            out[name + "/tvg_" + k] = v
        cases.append(name)
        print("%-22s matches %4d  config %d  inliers %4d  trials %s" % (name, len(m), tv.config, tv.num_inliers, list(tv.num_trials)))

    sc = synthetic.Scene(3, 256, seed=7, n_pool=320)
    ims = [sc.image(i) for i in range(3)]
    kps = [im[1].astype(np.float64) for im in ims]
    add("general_calibrated", ims[0][0], ims[1][0], kps[0], kps[1], 1, 11)
    add("general_uncalibrated", ims[0][0], ims[2][0], kps[0], kps[2], 0, 12)
    sp = synthetic.Scene(2, 256, seed=3, n_pool=320, planar=True)
    a, b = sp.image(0), sp.image(1)
    add("planar_calibrated", a[0], b[0], a[1].astype(np.float64), b[1].astype(np.float64), 1, 13)
    # two independently moving structures side by side -> EstimateMultiple finds both
    s1, s2 = synthetic.Scene(2, 160, seed=101, n_pool=200), synthetic.Scene(2, 160, seed=202, n_pool=200)
    a1, a2, b1, b2 = s1.image(0), s1.image(1), s2.image(0), s2.image(1)
    add("two_motions_multiple", np.concatenate([a1[0], b1[0]]), np.concatenate([a2[0], b2[0]]),
        np.concatenate([a1[1], b1[1]]).astype(np.float64), np.concatenate([a2[1], b2[1]]).astype(np.float64), 1, 14, multiple=1)
    out["cases"] = np.array(cases)
    path = os.path.join(ROOT, "tests", "golden", "pairs_v1.npz")
    if "--v1" in sys.argv:  # v1 is frozen; rewrite it only on purpose
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")
    main_v2(orc)


def cam_to_array(c):
    return np.array([c.model_id, c.has_prior_focal_length, c.width, c.height] + list(c.params), dtype=np.float64)


def main_v2(orc):
    """pairs_v2.npz (round 2): cases v1 lacks -- configurations PLANAR and PANORAMIC, the distorted camera models
    RADIAL / OPENCV / FULL_OPENCV (cameras stored per case), and a verification-only case at the benchmark's
    per-pair size (256 matches of two 4 096-feature images; descriptors omitted, matches stored as input)."""
    out, cases = {}, []

    def add(name, im1, im2, cam1, cam2, seed, with_desc=True, **optkw):
        opts = capi.default_two_view_options(**optkw)
        m = orc.match_sift_features_cpu(im1[0], im2[0])
        k1, k2 = im1[1].astype(np.float64), im2[1].astype(np.float64)
        tv, inl = orc.estimate_two_view_geometry(cam1, k1, cam2, k2, m, opts, seed)
        if with_desc:
            out[name + "/desc1"], out[name + "/desc2"] = im1[0], im2[0]
        out[name + "/kp1"], out[name + "/kp2"] = k1, k2
        out[name + "/cam1"], out[name + "/cam2"] = cam_to_array(cam1), cam_to_array(cam2)
        out[name + "/params"] = np.array([seed, opts.multiple_models, opts.min_num_inliers], dtype=np.int64)
        out[name + "/matches"], out[name + "/inlier_matches"] = m, inl
        for k, v in tvg_to_arrays(tv).items():
            out[name + "/tvg_" + k] = v
        cases.append(name)
        print("%-22s matches %4d  config %d  inliers %4d  trials %s" % (name, len(m), tv.config, tv.num_inliers, list(tv.num_trials)))
        return tv

    pin = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1)
    sp = synthetic.Scene(2, 320, seed=31, n_pool=400, planar=True, planar_depth=0.0)
    assert add("planar", sp.image(0), sp.image(1), pin, pin, 21).config == 4
    so = synthetic.Scene(2, 320, seed=32, n_pool=400, panoramic=True, kp_sigma=0.1)
    assert add("panoramic", so.image(0), so.image(1), pin, pin, 22).config == 5
    for name, mid, par in [("radial", 3, [800.0, 500.0, 375.0, 0.05, 0.01]),
                           ("opencv", 4, [800.0, 805.0, 500.0, 375.0, -0.12, 0.05, -0.001, 0.001]),
                           ("full_opencv", 6, [800.0, 805.0, 500.0, 375.0, -0.12, 0.05, -0.001, 0.001, 0.001, 0.02, -0.02, 0.001])]:
        sc = synthetic.Scene(2, 320, seed=40 + mid, n_pool=400, camera=(mid, par))
        assert add(name, sc.image(0), sc.image(1), capi.camera(mid, par, 1000, 750, True), pin, 23).config == 2
    sb = synthetic.Scene(2, 4096, seed=0)  # the bench scene's first pair
    assert add("bench_pair_4096", sb.image(0), sb.image(1), pin, pin, capi.pair_seed(0, 1, 0), with_desc=False).num_inliers > 120
    out["cases"] = np.array(cases)
    path = os.path.join(ROOT, "tests", "golden", "pairs_v2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
