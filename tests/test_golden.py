"""Pair-level golden vectors (tests/golden/pairs_v1.npz, written by tools/make_golden.py): inputs and outputs of
MatchSiftFeaturesCPU + TwoViewGeometry::Estimate / EstimateMultiple for a few small pairs.  The oracle must keep
reproducing them (a change of its defined arithmetic is a deliberate, documented event) and the HIP path must
produce the same: match indices, inlier matches, config, E / F / H and trial counts bit-exact, pose within 1e-6."""
import os

import numpy as np
import pytest

from dagsfm_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairs_v1.npz")


def _cases():
    g = np.load(GOLDEN)
    return g, [str(c) for c in g["cases"]]


def _check(g, name, m, tv, inl, pose_rtol):
    assert (m == g[name + "/matches"]).all()
    assert tv.config == int(g[name + "/tvg_config"]) and tv.num_inliers == int(g[name + "/tvg_num_inliers"])
    assert (inl == g[name + "/inlier_matches"]).all()
    assert list(tv.num_trials) == list(g[name + "/tvg_num_trials"]) and list(tv.num_models) == list(g[name + "/tvg_num_models"])
    for k in ("E", "F", "H"):
        assert (np.array(getattr(tv, k)) == g[name + "/tvg_" + k]).all(), (name, k)
    assert np.allclose(np.array(tv.qvec), g[name + "/tvg_qvec"], rtol=pose_rtol, atol=1e-12)
    assert np.allclose(np.array(tv.tvec), g[name + "/tvg_tvec"], rtol=pose_rtol, atol=1e-12)
    assert abs(tv.tri_angle - float(g[name + "/tvg_tri_angle"])) <= pose_rtol * max(abs(float(g[name + "/tvg_tri_angle"])), 1e-9)


def _setup(g, name):
    prior, seed, multiple, mni = [int(v) for v in g[name + "/params"]]
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior)
    opts = capi.default_two_view_options()
    opts.multiple_models = multiple
    opts.min_num_inliers = mni
    return cam, opts, seed


def test_oracle_reproduces_golden(oracle):
    g, cases = _cases()
    assert len(cases) >= 4
    for name in cases:
        cam, opts, seed = _setup(g, name)
        m = oracle.match_sift_features_cpu(g[name + "/desc1"], g[name + "/desc2"])
        tv, inl = oracle.estimate_two_view_geometry(cam, g[name + "/kp1"], cam, g[name + "/kp2"], m, opts, seed)
        _check(g, name, m, tv, inl, 1e-9)


@pytest.mark.gpu
def test_device_reproduces_golden(dsm):
    g, cases = _cases()
    for name in cases:
        cam, opts, seed = _setup(g, name)
        m = dsm.match_sift_features(g[name + "/desc1"], g[name + "/desc2"])
        tv, inl = dsm.estimate_two_view_geometry(cam, g[name + "/kp1"], cam, g[name + "/kp2"], m, opts, seed)
        _check(g, name, m, tv, inl, 1e-6)


# ---- pairs_v2.npz (round 2): PLANAR / PANORAMIC configurations, distorted camera models, one benchmark-size pair
GOLDEN2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pairs_v2.npz")


def _cam_from_array(a):
    return capi.camera(int(a[0]), list(a[4:]), int(a[2]), int(a[3]), bool(a[1]))


def _setup2(g, name):
    seed, multiple, mni = [int(v) for v in g[name + "/params"]]
    opts = capi.default_two_view_options(multiple_models=multiple, min_num_inliers=mni)
    return _cam_from_array(g[name + "/cam1"]), _cam_from_array(g[name + "/cam2"]), opts, seed


def test_oracle_reproduces_golden_v2(oracle):
    g = np.load(GOLDEN2)
    cases = [str(c) for c in g["cases"]]
    assert {"planar", "panoramic", "radial", "opencv", "full_opencv", "bench_pair_4096"} <= set(cases)
    assert int(g["planar/tvg_config"]) == 4 and int(g["panoramic/tvg_config"]) == 5
    for name in cases:
        cam1, cam2, opts, seed = _setup2(g, name)
        m = g[name + "/matches"]
        if name + "/desc1" in g:
            m = oracle.match_sift_features_cpu(g[name + "/desc1"], g[name + "/desc2"])
        tv, inl = oracle.estimate_two_view_geometry(cam1, g[name + "/kp1"], cam2, g[name + "/kp2"], m, opts, seed)
        _check(g, name, m, tv, inl, 1e-9)


@pytest.mark.gpu
def test_device_reproduces_golden_v2(dsm):
    g = np.load(GOLDEN2)
    for name in [str(c) for c in g["cases"]]:
        cam1, cam2, opts, seed = _setup2(g, name)
        m = g[name + "/matches"]
        if name + "/desc1" in g:
            m = dsm.match_sift_features(g[name + "/desc1"], g[name + "/desc2"])
        tv, inl = dsm.estimate_two_view_geometry(cam1, g[name + "/kp1"], cam2, g[name + "/kp2"], m, opts, seed)
        _check(g, name, m, tv, inl, 1e-6)
