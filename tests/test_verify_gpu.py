"""GPU parity of the HIP two-view verification (through the C-ABI) against the CPU oracle:
bit-exact config / inlier masks / model matrices / trial counts, pose within 1e-6 relative."""
import os

import numpy as np
import pytest

from dagsfm_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def tvg_equal(g, r, tag=""):
    assert g.config == r.config, (tag, g.config, r.config)
    assert g.num_inliers == r.num_inliers, (tag, g.num_inliers, r.num_inliers)
    assert g.num_matches == r.num_matches
    assert list(g.num_trials) == list(r.num_trials), (tag, list(g.num_trials), list(r.num_trials))
    assert list(g.num_models) == list(r.num_models), (tag, list(g.num_models), list(r.num_models))
    for name in ("E", "F", "H"):
        a, b = np.array(getattr(g, name)), np.array(getattr(r, name))
        assert (a == b).all() or (np.isnan(a) == np.isnan(b)).all() and np.allclose(a, b, rtol=0, atol=0, equal_nan=True), \
            (tag, name, a, b)
    assert np.allclose(np.array(g.qvec), np.array(r.qvec), rtol=1e-6, atol=1e-12), (tag, list(g.qvec), list(r.qvec))
    assert np.allclose(np.array(g.tvec), np.array(r.tvec), rtol=1e-6, atol=1e-12), (tag, list(g.tvec), list(r.tvec))
    assert abs(g.tri_angle - r.tri_angle) <= 1e-6 * max(abs(r.tri_angle), 1e-9), (tag, g.tri_angle, r.tri_angle)


def test_sampler_matches_libstdcxx(dsm, oracle):
    """Device MT19937 + Lemire + partial Fisher-Yates == std::mt19937 + std::uniform_int_distribution."""
    for seed, k, total, draws in [(0, 7, 50, 300), (5489, 5, 5, 10), (123456789, 4, 4096, 2000), (42, 1, 1, 5),
                                  (7, 7, 257, 1000), (4294967295, 4, 3, 0),
                                  (11, 4, 6, 700), (12, 7, 9, 700), (13, 5, 8, 700), (14, 7, 7, 50)]:  # partners inside the head, repeats
        if k > total or draws == 0:
            continue
        ref = oracle.sample_sequence(seed, k, total, draws)
        # 1 = the wave sampler of the product path (parallel regeneration / tempering / Lemire, serial swaps),
        # 0 = plain lane-0 loop, 2 = the wave sampler's serial replay path (taken after a Lemire rejection)
        assert (dsm.debug_sample_sequence(seed, k, total, draws) == ref).all(), (seed, k, total, draws, "product")
        for mode in ("1", "0", "2"):  # (a check-build switch: these calls go to the check library's context)
            os.environ["DSM_DEBUG_SAMPLER_MODE"] = mode
            try:
                assert (dsm.debug_sample_sequence(seed, k, total, draws) == ref).all(), (seed, k, total, draws, mode)
            finally:
                del os.environ["DSM_DEBUG_SAMPLER_MODE"]


def _scene_pair(scene, i, j, oracle):
    a, b = scene.image(i), scene.image(j)
    m = oracle.match_sift_features_cpu(a[0], b[0])
    return a[1].astype(np.float64), b[1].astype(np.float64), m


@pytest.mark.parametrize("prior", [0, 1])
def test_leaf_estimate_general_scene(dsm, oracle, prior):
    scene = synthetic.Scene(4, 1024, seed=11)
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior)
    opts = capi.default_two_view_options()
    for (i, j) in [(0, 1), (0, 2), (1, 3)]:
        p1, p2, m = _scene_pair(scene, i, j, oracle)
        for seed in (1, 77):
            ref, ref_inl = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, seed)
            got, got_inl = dsm.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, seed)
            tvg_equal(got, ref, (i, j, seed, prior))
            assert (got_inl == ref_inl).all()
            assert ref.num_inliers > 15


def test_leaf_estimate_planar_and_degenerate(dsm, oracle):
    scene = synthetic.Scene(3, 1024, seed=3, planar=True)
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True)
    camu = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, False)
    opts = capi.default_two_view_options()
    p1, p2, m = _scene_pair(scene, 0, 1, oracle)
    for c in (cam, camu):
        ref, ref_inl = oracle.estimate_two_view_geometry(c, p1, c, p2, m, opts, 5)
        got, got_inl = dsm.estimate_two_view_geometry(c, p1, c, p2, m, opts, 5)
        tvg_equal(got, ref, "planar")
        assert (got_inl == ref_inl).all()
    # too few matches -> DEGENERATE (two_view_geometry.cc:298-301)
    ref, _ = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m[:10], opts, 5)
    got, _ = dsm.estimate_two_view_geometry(cam, p1, cam, p2, m[:10], opts, 5)
    assert ref.config == 1
    tvg_equal(got, ref, "few")
    # pure outliers: random correspondences
    rng = np.random.default_rng(0)
    q1 = rng.uniform(0, 1000, (200, 2))
    q2 = rng.uniform(0, 1000, (200, 2))
    mm = np.stack([np.arange(200), rng.permutation(200)], axis=1).astype(np.uint32)
    for c in (cam, camu):
        ref, ref_inl = oracle.estimate_two_view_geometry(c, q1, c, q2, mm, opts, 9)
        got, got_inl = dsm.estimate_two_view_geometry(c, q1, c, q2, mm, opts, 9)
        tvg_equal(got, ref, "outliers")
        assert (got_inl == ref_inl).all()


def test_watermark_configuration(dsm, oracle):
    """Matches concentrated in the border that follow a pure translation -> WATERMARK (two_view_geometry.cc:491-555)."""
    rng = np.random.default_rng(1)
    n = 120
    p1 = np.c_[rng.uniform(5, 60, n), rng.uniform(5, 700, n)]
    p2 = p1 + np.array([3.0, -2.0]) + rng.normal(scale=0.2, size=(n, 2))
    m = np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.uint32)
    camu = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, False)
    opts = capi.default_two_view_options()
    ref, ref_inl = oracle.estimate_two_view_geometry(camu, p1, camu, p2, m, opts, 3)
    got, got_inl = dsm.estimate_two_view_geometry(camu, p1, camu, p2, m, opts, 3)
    assert ref.config == 7
    tvg_equal(got, ref, "watermark")
    assert (got_inl == ref_inl).all()


@pytest.mark.parametrize("schedule", ["default", "legacy", "lanes", "chunks"])
def test_watermark_suspects_among_ordinary_pairs(dsm, oracle, schedule, monkeypatch):
    """The translation estimator's ComputeNumTrials tables are built on demand (round 3; round 2 tabulated every sample
    count up front, O(n^2) entries): k_verify_final parks a pair whose watermark test needs a table that does not exist,
    the host builds it and sends exactly those pairs through the kernel again.  Here: a pair list through the stage
    calls in which some pairs are watermark suspects with DIFFERENT inlier counts (several tables, one second visit),
    twice on the same context (the second call finds its tables), on every schedule."""
    if schedule == "legacy":
        monkeypatch.setenv("DSM_VERIFY_LEGACY", "1")
    if schedule == "lanes":
        monkeypatch.setenv("DSM_VERIFY_LANES", "2")
    if schedule == "chunks":
        monkeypatch.setenv("DSM_VERIFY_CHUNK_PAIRS", "3")
    rng = np.random.default_rng(12)
    n_img, nk = 8, 400
    camu = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, False)
    scene = synthetic.Scene(n_img, nk, seed=71, n_pool=900)
    ims = [scene.image(i) for i in range(n_img)]
    kps = [im[1].astype(np.float64).copy() for im in ims]
    pairs = synthetic.exhaustive_pairs(n_img)
    matches = [oracle.match_sift_features_cpu(ims[int(i)][0], ims[int(j)][0]) for i, j in pairs]
    # watermark suspects: replace the keypoints the matches of three pairs refer to by a border cluster that moves
    # by a pure translation (disjoint images, so that no other pair of the list is disturbed: pairs (0,1) (2,3) (4,5))
    wm_pairs = [k for k, (i, j) in enumerate(pairs) if (int(i), int(j)) in ((0, 1), (2, 3), (4, 5))]
    for q, k in enumerate(wm_pairs):
        i, j = int(pairs[k][0]), int(pairs[k][1])
        n = 60 + 25 * q
        idx = np.arange(n)
        kps[i][idx] = np.c_[rng.uniform(5, 60, n), rng.uniform(5, 700, n)]
        kps[j][idx] = kps[i][idx] + np.array([3.0, -2.0]) + rng.normal(scale=0.2, size=(n, 2))
        matches[k] = np.stack([idx, idx], axis=1).astype(np.uint32)
    opts = capi.default_two_view_options()
    dsm.set_images([im[0] for im in ims], [k.astype(np.float32) for k in kps], [camu] * n_img)
    kps = [k.astype(np.float32).astype(np.float64) for k in kps]
    for rep in range(2):
        dsm.set_matches(pairs, matches)
        dsm.verify_pairs(opts, user_seed=4, stage_filter=False)
        tvgs = dsm.two_view_geometries()
        ioffs, im = dsm.inlier_matches()
        n_wm = 0
        for k, (i, j) in enumerate(pairs):
            ref, ref_inl = oracle.estimate_two_view_geometry(camu, kps[int(i)], camu, kps[int(j)], matches[k], opts,
                                                             capi.pair_seed(int(i), int(j), 4))
            tvg_equal(tvgs[k], ref, (rep, int(i), int(j)))
            assert (im[int(ioffs[k]):int(ioffs[k + 1])] == ref_inl).all()
            n_wm += ref.config == 7
        assert n_wm == 3


@pytest.mark.parametrize("prior,sampler_serial,legacy", [(0, False, False), (1, False, False), (1, True, False), (1, False, True),
                                                         (1, False, "inline_lo"), (0, False, "inline_lo"), (1, False, "chunks"),
                                                         (1, False, "batched_lo"), (0, False, "batched_lo"),
                                                         (1, False, "lanes"), (0, False, "lanes_inline"),
                                                         (1, False, "batched_tail"), (0, False, "batched_tail"),
                                                         (1, False, "batched_tail_all"), (0, False, "batched_tail_all"),
                                                         (1, False, "batched_tail_inline"), (0, False, "batched_tail_inline"),
                                                         (1, False, "item_mode"), (0, False, "item_mode")])
def test_stage_match_and_verify_many_pairs(dsm, oracle, prior, sampler_serial, legacy, monkeypatch):
    """dsm_match_pairs + dsm_verify_pairs over an exhaustive pair list == oracle matcher + oracle verifier
    with SiftFeatureMatcher::Match's post-filter (matching.cc:824-831).  sampler_serial forces the sampler's
    rarely taken serial replay path (a Lemire rejection) for every round; legacy runs the single-kernel-per-family
    schedule (DSM_VERIFY_LEGACY)."""
    if sampler_serial:
        monkeypatch.setenv("DSM_SAMPLER_SERIAL", "1")
    # DSM_LO_TAIL: queue length at which the batched schedule finishes a round inline (default 2 048 = always on lists
    # this short); 0 keeps every local optimisation in the batched kernels, "batched_tail" mixes the two
    monkeypatch.setenv("DSM_LO_TAIL", {"batched_tail": "6", "batched_tail_all": "100000", "batched_tail_inline": "100000"}.get(legacy, "0"))
    monkeypatch.setenv("DSM_LO_TAIL_MODE", "inline" if legacy == "batched_tail_inline" else "items")  # the tail as an item pass (default) or inline
    monkeypatch.setenv("DSM_VERIFY_ITEM_MODE", "1" if legacy == "item_mode" else "0")  # every round as item passes from the start
    if legacy == "chunks":  # several chunks of the pair list (one chunk is the rule on a 288 GB device)
        monkeypatch.setenv("DSM_VERIFY_CHUNK_PAIRS", "5")
        monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "0")
    elif legacy in ("lanes", "lanes_inline"):  # three concurrent lanes (host threads + streams), several chunks each
        monkeypatch.setenv("DSM_VERIFY_LANES", "3")
        monkeypatch.setenv("DSM_VERIFY_CHUNK_PAIRS", "4")
        monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "0" if legacy == "lanes" else "1")
    elif legacy in ("batched_lo", "batched_tail", "batched_tail_all", "batched_tail_inline", "item_mode"):  # the schedule long pair lists get (short ones default to the inline form)
        monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "0")
    elif legacy == "inline_lo":  # phase-split pipeline with the local optimisation inline in the replay (round-1 schedule)
        monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "1")
    elif legacy:  # the first schedule of round 1: one k_ransac kernel per family, wave per pair
        monkeypatch.setenv("DSM_VERIFY_LEGACY", "1")
    n_img = 7
    scene = synthetic.Scene(n_img, 768, seed=21, n_pool=2048)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior) for _ in range(n_img)]
    dsm.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    pairs = synthetic.exhaustive_pairs(n_img)
    dsm.match_pairs(pairs)
    opts = capi.default_two_view_options()
    dsm.verify_pairs(opts, user_seed=99, stage_filter=True)
    offs, m = dsm.matches()
    tvgs = dsm.two_view_geometries()
    ioffs, im = dsm.inlier_matches()
    n_verified = 0
    for k, (i, j) in enumerate(pairs):
        mk = m[int(offs[k]):int(offs[k + 1])]
        ref_m = oracle.match_sift_features_cpu(ims[i][0], ims[j][0])
        assert (mk == ref_m).all()
        seed = capi.pair_seed(int(i), int(j), 99)
        ref, ref_inl = oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j],
                                                         ims[j][1].astype(np.float64), ref_m, opts, seed)
        got = tvgs[k]
        got_inl = im[int(ioffs[k]):int(ioffs[k + 1])]
        if ref.num_inliers < opts.min_num_inliers:
            assert got.config == 0 and got.num_inliers == 0 and len(got_inl) == 0
        else:
            tvg_equal(got, ref, (i, j))
            assert (got_inl == ref_inl).all()
            n_verified += 1
    assert n_verified >= 10
    assert dsm.verify_kernel_time() > 0


@pytest.mark.parametrize("form", ["product", "hyp_pair_grid", "lo_prepare_wave", "elu_lds", "replay_legacy"])
@pytest.mark.parametrize("planar,outlier_frac", [(False, 0.5), (True, 0.2)])
def test_round6_forms_on_many_rounds_and_planar_scenes(dsm, oracle, form, planar, outlier_frac, monkeypatch):
    """Round 6's scheduling forms against the oracle where they matter: a 0.25 inlier ratio (E / F run MANY rounds: from a pair's
    second round on the lane-per-hypothesis solvers take the hypotheses of all pairs from k_sample's list, verify_kernels.hip
    hyp_of_lane) and a planar scene (the local optimisations of H have hundreds of inliers: k_lo_prepare_reg<H, 3>; those of a general
    scene a handful: <H, 1>).  `product` is the product library; the other forms are the check build's older ones of the same results
    (DSM_HYP_GRID=pair: the (pair, 64 trials) grid in every round; DSM_LO_PREPARE_WAVE: every design matrix through the general kernel
    k_lo_prepare; DSM_ELU_LDS: the 5-point elimination in LDS; DSM_REPLAY_LEGACY: the replay scans of rounds 2 - 5), batched schedule,
    two lanes and three chunks each, so that list segments, chunk boundaries and lanes all occur."""
    env = {"hyp_pair_grid": ("DSM_HYP_GRID", "pair"), "lo_prepare_wave": ("DSM_LO_PREPARE_WAVE", "1"), "elu_lds": ("DSM_ELU_LDS", "1"),
           "replay_legacy": ("DSM_REPLAY_LEGACY", "1")}.get(form)
    if env:
        monkeypatch.setenv(*env)
    monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "0")
    monkeypatch.setenv("DSM_VERIFY_LANES", "2")
    monkeypatch.setenv("DSM_VERIFY_CHUNK_PAIRS", "6")
    monkeypatch.setenv("DSM_LO_TAIL", "3")
    n_img = 8
    scene = synthetic.Scene(n_img, 768, seed=61 + planar, n_pool=2048, planar=planar, outlier_frac=outlier_frac)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(n_img)]
    ctx = dsm  # (the fixture hands out the check build's context while a check-only switch is set)
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    pairs = synthetic.exhaustive_pairs(n_img)
    ctx.match_pairs(pairs)
    opts = capi.default_two_view_options()
    ctx.verify_pairs(opts, user_seed=7, stage_filter=False)
    offs, m = ctx.matches()
    tvgs = ctx.two_view_geometries()
    ioffs, im = ctx.inlier_matches()
    n_geo = n_many = 0
    for k, (i, j) in enumerate(pairs):
        mk = m[int(offs[k]):int(offs[k + 1])]
        ref, ref_inl = oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j], ims[j][1].astype(np.float64), mk, opts,
                                                         capi.pair_seed(int(i), int(j), 7))
        tvg_equal(tvgs[k], ref, (form, planar, i, j))
        assert (im[int(ioffs[k]):int(ioffs[k + 1])] == ref_inl).all()
        n_geo += ref.config > 1
        n_many += ref.num_trials[0] > 64 or ref.num_trials[1] > 128   # E / F went beyond their first round
    assert n_geo >= 10
    if not planar:
        assert n_many >= 10, "the workload must take E / F beyond their first round"


@pytest.mark.parametrize("schedule", ["default", "batched"])
def test_radial_camera_and_small_lo_systems(dsm, oracle, schedule, monkeypatch):
    """SIMPLE_RADIAL cameras (iterative undistortion) and tiny inlier sets (6..9-row LO systems).  "batched": the
    local optimisations through the batched kernels, where such a system is one of the few that the general kernels
    (k_lo_prepare, k_lo_jacobi) still serve next to the register-resident ones."""
    if schedule == "batched":
        monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "0")
        monkeypatch.setenv("DSM_LO_TAIL", "0")
    scene = synthetic.Scene(3, 512, seed=8)
    cam = capi.Camera(model_id=2, has_prior_focal_length=1, width=1000, height=750)
    cam.params[0], cam.params[1], cam.params[2], cam.params[3] = 800.0, 500.0, 375.0, 0.05
    opts = capi.default_two_view_options()
    p1, p2, m = _scene_pair(scene, 0, 1, oracle)
    ref, ref_inl = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, 2)
    got, got_inl = dsm.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, 2)
    tvg_equal(got, ref, "radial")
    assert (got_inl == ref_inl).all()
    # 16..24 matches with only a handful of true correspondences
    rng = np.random.default_rng(4)
    camp = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True)
    for n_true, n_tot in [(8, 16), (12, 20), (6, 24)]:
        mm = m[:n_true]
        extra = np.stack([rng.choice(len(p1), n_tot - n_true, replace=False),
                          rng.choice(len(p2), n_tot - n_true, replace=False)], axis=1).astype(np.uint32)
        mx = np.concatenate([mm, extra])
        o2 = capi.default_two_view_options(min_num_inliers=6)
        for seed in (1, 2, 3):
            ref, ref_inl = oracle.estimate_two_view_geometry(camp, p1, camp, p2, mx, o2, seed)
            got, got_inl = dsm.estimate_two_view_geometry(camp, p1, camp, p2, mx, o2, seed)
            tvg_equal(got, ref, ("small", n_true, n_tot, seed))
            assert (got_inl == ref_inl).all()


def _two_motion_problem(oracle):
    """Correspondences of two independently moving structures: the matches of two different scene pairs put side
    by side (indices of the second set shifted), so that a second Estimate pass over the outliers of the first
    finds another geometry."""
    sa = synthetic.Scene(2, 640, seed=101, n_pool=900)
    sb = synthetic.Scene(2, 640, seed=202, n_pool=900)
    a1, a2, ma = _scene_pair(sa, 0, 1, oracle)
    b1, b2, mb = _scene_pair(sb, 0, 1, oracle)
    p1 = np.concatenate([a1, b1])
    p2 = np.concatenate([a2, b2])
    m = np.concatenate([ma, mb + np.array([len(a1), len(a2)], dtype=np.uint32)]).astype(np.uint32)
    return p1, p2, m, len(ma), len(mb)


@pytest.mark.parametrize("prior", [0, 1])
def test_estimate_multiple_leaf(dsm, oracle, prior):
    """TwoViewGeometry::EstimateMultiple (two_view_geometry.cc:128-167): repeated passes over the remaining
    matches on one generator stream; one geometry -> that geometry, several -> MULTIPLE + all inlier matches."""
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior)
    opts = capi.default_two_view_options()
    opts.multiple_models = 1
    p1, p2, m, na, nb = _two_motion_problem(oracle)
    assert na > 60 and nb > 60
    for seed in (4, 19):
        ref, ref_inl = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, seed)
        got, got_inl = dsm.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, seed)
        assert ref.config == 8, ref.config  # MULTIPLE
        tvg_equal(got, ref, ("multiple", prior, seed))
        assert (got_inl == ref_inl).all()
    # a single rigid scene: the second pass is DEGENERATE -> the first geometry, trial counters of both passes
    scene = synthetic.Scene(2, 1024, seed=11)
    q1, q2, mm = _scene_pair(scene, 0, 1, oracle)
    single = capi.default_two_view_options()
    ref1, _ = oracle.estimate_two_view_geometry(cam, q1, cam, q2, mm, single, 7)
    ref, ref_inl = oracle.estimate_two_view_geometry(cam, q1, cam, q2, mm, opts, 7)
    got, got_inl = dsm.estimate_two_view_geometry(cam, q1, cam, q2, mm, opts, 7)
    assert ref.config == ref1.config and ref.num_inliers == ref1.num_inliers
    assert sum(ref.num_trials) > sum(ref1.num_trials)
    tvg_equal(got, ref, ("single", prior))
    assert (got_inl == ref_inl).all()


def test_estimate_multiple_stage(dsm, oracle):
    """multiple_models through dsm_verify_pairs over a pair list (pairs finish after different numbers of passes)."""
    n_img = 5
    scene = synthetic.Scene(n_img, 768, seed=33, n_pool=2048)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(n_img)]
    dsm.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    pairs = synthetic.exhaustive_pairs(n_img)
    dsm.match_pairs(pairs)
    opts = capi.default_two_view_options()
    opts.multiple_models = 1
    opts.min_num_inliers = 8  # lets second passes find small structures among the leftovers
    dsm.verify_pairs(opts, user_seed=5, stage_filter=True)
    offs, m = dsm.matches()
    tvgs = dsm.two_view_geometries()
    ioffs, im = dsm.inlier_matches()
    configs = set()
    for k, (i, j) in enumerate(pairs):
        mk = m[int(offs[k]):int(offs[k + 1])]
        seed = capi.pair_seed(int(i), int(j), 5)
        ref, ref_inl = oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j],
                                                         ims[j][1].astype(np.float64), mk, opts, seed)
        got = tvgs[k]
        got_inl = im[int(ioffs[k]):int(ioffs[k + 1])]
        if ref.num_inliers < opts.min_num_inliers:
            assert got.config == 0 and got.num_inliers == 0 and len(got_inl) == 0
        else:
            tvg_equal(got, ref, (i, j))
            assert (got_inl == ref_inl).all()
        configs.add(ref.config)
    assert len(configs) >= 1


def test_baseline_config1_size(dsm, oracle):
    """BASELINE.json configs[0] shape: 50 images x 1 024 features, exhaustive (1 225 pairs), cameras without focal
    prior (F + H path).  Matching is compared with the oracle on every 5th pair, verification on every 25th;
    over ALL pairs the size-independent properties: ascending unique idx1, unique idx2, inlier matches an ordered
    subsequence of the matches, config / inlier-count consistency."""
    n_img = 50
    scene = synthetic.Scene(n_img, 1024, seed=42)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 0) for _ in range(n_img)]
    dsm.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    pairs = synthetic.exhaustive_pairs(n_img)
    assert len(pairs) == 1225
    dsm.match_pairs(pairs)
    opts = capi.default_two_view_options()
    dsm.verify_pairs(opts, user_seed=1, stage_filter=True)
    offs, m = dsm.matches()
    tvgs = dsm.two_view_geometries()
    ioffs, im = dsm.inlier_matches()
    n_geo = 0
    for k, (i, j) in enumerate(pairs):
        mk = m[int(offs[k]):int(offs[k + 1])]
        ik = im[int(ioffs[k]):int(ioffs[k + 1])]
        if len(mk) > 1:
            assert (np.diff(mk[:, 0].astype(np.int64)) > 0).all()
        assert len(np.unique(mk[:, 1])) == len(mk)
        assert tvgs[k].num_inliers == len(ik) and tvgs[k].num_matches == len(mk)
        if len(ik):
            pos = np.searchsorted(mk[:, 0], ik[:, 0])
            assert (mk[pos] == ik).all() and (np.diff(pos) > 0).all()
            assert tvgs[k].config in (3, 4, 5, 6, 7) and len(ik) >= opts.min_num_inliers
            n_geo += 1
        else:
            assert tvgs[k].config == 0
        if k % 5 == 0:
            ref_m = oracle.match_sift_features_cpu(ims[i][0], ims[j][0])
            assert (mk == ref_m).all(), (i, j)
            if k % 25 == 0:
                ref, ref_inl = oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j],
                                                                 ims[j][1].astype(np.float64), ref_m, opts,
                                                                 capi.pair_seed(int(i), int(j), 1))
                if ref.num_inliers < opts.min_num_inliers:
                    assert tvgs[k].config == 0
                else:
                    tvg_equal(tvgs[k], ref, (i, j))
                    assert (ik == ref_inl).all()
    assert n_geo > 100


@pytest.mark.parametrize("prior,planar,cross", [(1, False, True), (0, False, True), (1, True, True), (1, False, False)])
def test_guided_matching_stage(dsm, oracle, prior, planar, cross):
    """SiftMatchingOptions::guided_matching: verify, then MatchGuidedSiftFeaturesCPU (sift.cc:824-875) for every pair
    with enough inliers and an F- or H-type configuration replaces the inlier matches (matching.cc:441-470), then
    Match()'s post-filter.  Compared with the oracle pair by pair."""
    n_img = 5
    scene = synthetic.Scene(n_img, 640, seed=51 + prior, n_pool=1500, planar=planar)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior) for _ in range(n_img)]
    dsm.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    pairs = synthetic.exhaustive_pairs(n_img)
    mo = capi.default_match_options()
    mo.cross_check = 1 if cross else 0
    dsm.match_pairs(pairs, mo)
    opts = capi.default_two_view_options()
    dsm.verify_pairs(opts, user_seed=8, stage_filter=False)
    pre = dsm.two_view_geometries()
    pre_cfg = [t.config for t in pre]
    dsm.guided_match_pairs(mo, opts, stage_filter=True)
    offs, m = dsm.matches()
    tvgs = dsm.two_view_geometries()
    ioffs, im = dsm.inlier_matches()
    n_guided = 0
    for k, (i, j) in enumerate(pairs):
        mk = m[int(offs[k]):int(offs[k + 1])]
        ref, ref_inl = oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j],
                                                         ims[j][1].astype(np.float64), mk, opts, capi.pair_seed(int(i), int(j), 8))
        assert ref.config == pre_cfg[k]
        exp = ref_inl
        if ref.num_inliers >= opts.min_num_inliers:
            g = oracle.match_guided_sift_features_cpu(ims[i][1], ims[j][1], ims[i][0], ims[j][0], ref, max_error=opts.max_error,
                                                      cross_check=cross)
            if g is not None:
                exp = g
                n_guided += 1
        got = tvgs[k]
        got_inl = im[int(ioffs[k]):int(ioffs[k + 1])]
        if len(exp) < opts.min_num_inliers:
            assert got.config == 0 and got.num_inliers == 0 and len(got_inl) == 0
        else:
            assert got.config == ref.config and got.num_inliers == len(exp)
            assert (got_inl == exp).all(), (i, j)
            for name in ("E", "F", "H"):
                assert (np.array(getattr(got, name)) == np.array(getattr(ref, name))).all()
    assert n_guided >= 5


def test_differential_fuzz_small_pairs(dsm):
    """tools/fuzz_verify.py as a regression test: three option sets x 700 small seeded pairs of every structure the
    generator knows (general / planar / rotation / watermark / collinear / repeated points / outliers / a handful of
    matches; all camera models) through the stage calls on ONE context, every record and inlier list against the
    oracle.  Seed 1, batches 1 and 2 are the ones that found (round 3): a family that never runs (fewer matches than
    its minimal sample with min_num_inliers = 0) left the PREVIOUS call's report in the pair's slot; a watermark suspect
    parked for its translation table lost the "resume from the snapshot" mark of an early-stopped H family, so its
    translation RANSAC drew from the wrong place of the stream."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_verify
    msgs = []
    total, bad, configs = fuzz_verify.run_fuzz(dsm, 3, 700, 1, min(32, os.cpu_count() or 4), log=msgs.append)
    assert bad == 0, "\n".join(msgs)
    assert total == 2100 and len(configs) >= 6


def test_estimate_multiple_ends_when_a_pass_removes_nothing(dsm, oracle):
    """min_num_inliers = 0 with fewer than 7 matches: F fails, H succeeds, the inlier list (F's mask) is empty, so the
    pass is not DEGENERATE and removes no match -- TwoViewGeometry::EstimateMultiple (two_view_geometry.cc:128-167) would
    repeat it indefinitely.  Product and oracle record the geometry and stop; the call returns and both agree."""
    rng = np.random.default_rng(5)
    camu = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, False)
    opts = capi.default_two_view_options(min_num_inliers=0, multiple_models=1, multiple_ignore_watermark=0)
    for n in (4, 5, 6):
        p1 = rng.uniform(100, 900, (n, 2))
        H = np.array([[1.02, 0.01, 5.0], [-0.01, 0.98, -3.0], [1e-5, 2e-5, 1.0]])
        q = np.c_[p1, np.ones(n)] @ H.T
        p2 = q[:, :2] / q[:, 2:]
        m = np.stack([np.arange(n), np.arange(n)], axis=1).astype(np.uint32)
        ref, ref_inl = oracle.estimate_two_view_geometry(camu, p1, camu, p2, m, opts, 3)
        got, got_inl = dsm.estimate_two_view_geometry(camu, p1, camu, p2, m, opts, 3)
        tvg_equal(got, ref, ("no progress", n))
        assert (got_inl == ref_inl).all()
        assert ref.config != 1 and ref.num_inliers == 0


def test_differential_fuzz_stage_calls(dsm):
    """tools/fuzz_stage.py as a regression test: 30 small seeded scenes through set_images -> match_pairs -> verify_pairs
    -> guided_match_pairs on ONE context, matches / records / guided inlier lists / post-filter against the oracle.
    Seed 1 is the sequence that found (round 3) a hang: scene 11 is the first pair list longer than any before it, most
    of its pairs have fewer matches than min_num_inliers, and k_verify_prep left the family states of such pairs
    unwritten -- in the freshly grown buffer k_sample took whatever was there for an active pair."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_stage
    msgs = []
    total, bad, stats = fuzz_stage.run_fuzz(dsm, 30, 1, min(32, os.cpu_count() or 4), log=msgs.append)
    assert bad == 0, "\n".join(msgs)
    assert total > 200 and stats["guided"] > 30


@pytest.mark.parametrize("prefilter", ["1", "0", "9", "17", "33"])  # 1: the product (H: packed-f32 first stage + FP64 list); 17: the pure FP64 k_prescore<H> of round 4; 9: its products on the FP64 matrix pipe (both check build)
def test_bound_and_exact_scoring_regimes(dsm, oracle, prefilter, monkeypatch):
    """Round 4: F / H / E scoring as bound + exact (k_prescore, k_score_needed).  The bounds switch
    themselves off where their margins do not hold -- a threshold below 2^-6 px^2, coordinates beyond 2^14 px -- and every
    regime must equal the oracle, with the plain k_score (DSM_SCORE_PREFILTER=0) as well: thresholds from 0.1 px to 60 px,
    keypoints scaled by 40 (40 000 px wide images), calibrated and not, a planar scene (most homographies are near-ties)."""
    monkeypatch.setenv("DSM_SCORE_PREFILTER", prefilter)
    cases = []
    for planar in (False, True):
        scene = synthetic.Scene(3, 1024, seed=21 + planar, planar=planar)
        for scale, max_error in ((1.0, 0.1), (1.0, 0.3), (1.0, 4.0), (1.0, 60.0), (40.0, 160.0), (40.0, 8.0)):
            p1, p2, m = _scene_pair(scene, 0, 1 + planar, oracle)
            cases.append((planar, scale, max_error, p1 * scale, p2 * scale, m))
    n_geo = 0
    for planar, scale, max_error, p1, p2, m in cases:
        for prior in (0, 1):
            cam = capi.simple_pinhole(800.0 * scale, 500.0 * scale, 375.0 * scale, int(1000 * scale), int(750 * scale), prior)
            opts = capi.default_two_view_options(max_error=max_error)
            ref, ref_inl = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, 13)
            got, got_inl = dsm.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, 13)
            tvg_equal(got, ref, (planar, scale, max_error, prior))
            assert (got_inl == ref_inl).all()
            n_geo += ref.config > 1
    assert n_geo >= 12


def test_scoring_bounds_hold_on_every_slot(oracle, monkeypatch):
    """DSM_SCORE_PREFILTER=check: every (model, pair) slot is scored exactly AND its exact count is held against the bound
    step's [lower, upper] -- a violated bound that happened not to change a decision would go unnoticed by
    the parity tests; this counts them.  Scenes: general and planar, 0.64 and 0.25 inlier ratios, calibrated and not, a tight
    and a loose threshold; plus the fuzz generator's structures (collinear, repeated, pure outliers, all camera models).
    Counter [14] (violations) must be 0 and [15] (slots the filter skips) must be most of them on an ordinary scene."""
    import importlib.util
    import sys
    monkeypatch.setenv("DSM_SCORE_PREFILTER", "check")
    ctx = capi.Context(0)  # (the check build: a cross-check switch is in the environment)
    assert ctx.check
    skipped_ordinary = None
    for planar, outlier_frac, prior, max_error in ((False, 0.2, 1, 4.0), (True, 0.2, 1, 4.0), (False, 0.5, 1, 4.0), (False, 0.2, 0, 1.0),
                                                  (False, 0.2, 1, 12.0)):
        n_img = 7
        scene = synthetic.Scene(n_img, 1536, seed=31, planar=planar, outlier_frac=outlier_frac)
        ims = [scene.image(i) for i in range(n_img)]
        cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, prior) for _ in range(n_img)]
        ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
        pairs = synthetic.exhaustive_pairs(n_img)
        ctx.match_pairs(pairs)
        ctx.verify_pairs(capi.default_two_view_options(max_error=max_error), user_seed=3, stage_filter=True)
        c = ctx.debug_verify_counters()
        assert c[14] == 0, (planar, outlier_frac, prior, max_error, int(c[14]))
        if not planar and outlier_frac == 0.2 and prior == 1 and max_error == 4.0:
            skipped_ordinary = int(c[15])
        # and the results of the checking schedule are the oracle's
        tv = ctx.two_view_geometries()
        offs, m = ctx.matches()
        i, j = pairs[0]
        ref, _ = oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j], ims[j][1].astype(np.float64),
                                                   m[int(offs[0]):int(offs[1])], capi.default_two_view_options(max_error=max_error),
                                                   capi.pair_seed(int(i), int(j), 3))
        tvg_equal(tv[0], ref, (planar, outlier_frac, prior, max_error))
    assert skipped_ordinary is not None and skipped_ordinary > 21 * 1000  # > 1 000 of ~2 000+ slots per pair are never scored exactly
    # the fuzz generator's pairs (tools/fuzz_verify.py): every structure, every camera model, through the stage calls
    spec = importlib.util.spec_from_file_location("fuzz_verify", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_verify.py"))
    fz = importlib.util.module_from_spec(spec)
    sys.modules["fuzz_verify"] = fz
    spec.loader.exec_module(fz)
    problems = [fz.make_pair(9000 + k) for k in range(300)]
    descs, kps, cams, pairs, matches = [], [], [], [], []
    for k, (c1, c2, kp1, kp2, mt, kind) in enumerate(problems):
        for c, kp in ((c1, kp1), (c2, kp2)):
            descs.append(np.zeros((len(kp), 128), np.uint8))
            kps.append(kp)
            cams.append(capi.camera(c[0], c[1], fz.W, fz.H, c[2]))
        pairs.append((2 * k, 2 * k + 1))
        matches.append(mt)
    ctx.set_images(descs, kps, cams)
    ctx.set_matches(np.array(pairs, np.uint32), matches)
    for max_error in (4.0, 0.7):
        ctx.verify_pairs(capi.default_two_view_options(max_error=max_error), user_seed=1, stage_filter=False)
        assert ctx.debug_verify_counters()[14] == 0


def test_debug_options_are_per_context_and_checked():
    """dsm_set_debug_option (round 4): the library reads no environment; a switch belongs to ONE context, an unknown key is an
    error, NULL removes a key.  (The Python binding forwards DSM_* variables of the process through this call.)"""
    a, b = capi.Context(0, check=True), capi.Context(0, check=False)  # a: the check build (cross-check switches), b: the product
    assert len(capi.PRODUCT_OPTION_KEYS) <= 8
    for key in capi.PRODUCT_OPTION_KEYS:  # the product knows its scheduling knobs ...
        b.set_debug_option(key, "1")
        b.set_debug_option(key, None)
    for key in capi.CHECK_OPTION_KEYS:  # ... and refuses every cross-check switch; the check build takes both kinds
        with pytest.raises(capi.DsmError):
            b.set_debug_option(key, "1")
        a.set_debug_option(key, "1")
        a.set_debug_option(key, None)
    a.set_debug_option("DSM_VERIFY_LANES", "1")
    assert a._debug.get("DSM_VERIFY_LANES") == "1" and "DSM_VERIFY_LANES" not in b._debug
    with pytest.raises(capi.DsmError):
        a.set_debug_option("DSM_NO_SUCH_SWITCH", "1")
    a.set_debug_option("DSM_VERIFY_LANES", None)
    assert "DSM_VERIFY_LANES" not in a._debug
    # the two contexts give the same results whatever one of them was told
    scene = synthetic.Scene(3, 512, seed=5)
    ims = [scene.image(i) for i in range(3)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True) for _ in range(3)]
    a.set_debug_option("DSM_VERIFY_INLINE_LO", "0")
    a.set_debug_option("DSM_SCORE_PREFILTER", "0")
    recs = []
    for c in (a, b):
        c.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
        c.match_pairs(synthetic.exhaustive_pairs(3))
        c.verify_pairs(capi.default_two_view_options(), user_seed=7, stage_filter=False)
        recs.append([bytes(t) for t in c.two_view_geometries()])
    assert recs[0] == recs[1]
