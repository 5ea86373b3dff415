"""Oracle parity at the shapes bench.py measures ("What's weak" 1): the small-scene tests of
test_verify_gpu.py never reach 256 matches / ~165 inliers per pair, n > VP_LDS_PTS (the global-memory scoring path),
tall-QR columns with >= 32 rows per lane, 8 192-feature images, or fixed-trial options.  These do, each through the
C-ABI stage calls (dsm_set_images / dsm_match_pairs / dsm_verify_pairs) against the CPU oracle on the same inputs:
match indices, inlier matches, config, E / F / H, trial and model counts bit-exact; qvec / tvec / tri_angle 1e-6.

The oracle calls run on a thread pool (ctypes releases the GIL; the scalar matcher needs ~1 s per 4 096 x 4 096 pair)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from dagsfm_amd import capi, synthetic
from tests.test_verify_gpu import tvg_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["batched_lo", "batched_tail", "batched_tail_all", "batched_tail_inline", "item_mode", "inline_lo"])
def lo_schedule(request, monkeypatch):
    """The schedules of the local optimisation: the batched kernels that long pair lists (bench.py) run -- pure
    (DSM_LO_TAIL=0: every queued pair goes through k_lo_prepare* / k_lo_jacobi*) and with the inline tail (a queue of
    <= 4 pairs finishes its round in k_replay_lo<TAIL>; the default of 512 would swallow these short lists whole) --
    and the inline form that the library picks by itself for lists as short as these tests'."""
    monkeypatch.setenv("DSM_VERIFY_INLINE_LO", "1" if request.param == "inline_lo" else "0")
    # batched_tail_all: every pair goes to the inline finish right after its first suspension (the configuration that
    # exposed a register-spill miscompile of k_replay_lo<TAIL> in round 2: tools/bisect_schedules.py)
    # The tail itself has two forms: an item pass (k_items_* + the batched kernels over jobs + the lookup replay: the product path, round 3) and
    # the inline finish of round 2 (batched_tail_inline: DSM_LO_TAIL_MODE=inline).
    monkeypatch.setenv("DSM_LO_TAIL", {"batched_tail": "4", "batched_tail_all": "100000", "batched_tail_inline": "100000"}.get(request.param, "0"))
    monkeypatch.setenv("DSM_LO_TAIL_MODE", "inline" if request.param == "batched_tail_inline" else "items")
    # item_mode: every round as item passes from the start (what short pair lists get by default)
    monkeypatch.setenv("DSM_VERIFY_ITEM_MODE", "1" if request.param == "item_mode" else "0")
    return request.param


def _workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 32))


def _oracle_matches(oracle, ims, pairs):
    with ThreadPoolExecutor(_workers()) as ex:
        return list(ex.map(lambda ij: oracle.match_sift_features_cpu(ims[ij[0]][0], ims[ij[1]][0]), [(int(i), int(j)) for i, j in pairs]))


def _oracle_verify(oracle, ims, cams, pairs, matches, opts, user_seed):
    def one(k):
        i, j = int(pairs[k][0]), int(pairs[k][1])
        return oracle.estimate_two_view_geometry(cams[i], ims[i][1].astype(np.float64), cams[j], ims[j][1].astype(np.float64),
                                                 matches[k], opts, capi.pair_seed(i, j, user_seed))
    with ThreadPoolExecutor(_workers()) as ex:
        return list(ex.map(one, range(len(pairs))))


_ORACLE_CACHE = {}   # the four LO schedules of the fixture compare against the same oracle results


def _compare_stage(dsm, oracle, ims, cams, pairs, opts, user_seed, ref_matches=None, stage_filter=True, cache_key=None):
    dsm.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    dsm.match_pairs(pairs)
    dsm.verify_pairs(opts, user_seed=user_seed, stage_filter=stage_filter)
    offs, m = dsm.matches()
    tvgs = dsm.two_view_geometries()
    ioffs, im = dsm.inlier_matches()
    if cache_key is not None and cache_key in _ORACLE_CACHE:
        ref_matches, refs = _ORACLE_CACHE[cache_key]
    else:
        if ref_matches is None:
            ref_matches = _oracle_matches(oracle, ims, pairs)
        refs = None
    for k in range(len(pairs)):
        assert (m[int(offs[k]):int(offs[k + 1])] == ref_matches[k]).all(), ("matches", tuple(pairs[k]))
    if refs is None:
        refs = _oracle_verify(oracle, ims, cams, pairs, ref_matches, opts, user_seed)
    if cache_key is not None:
        _ORACLE_CACHE[cache_key] = (ref_matches, refs)
    configs, n_ok = {}, 0
    for k, (ref, ref_inl) in enumerate(refs):
        got, got_inl = tvgs[k], im[int(ioffs[k]):int(ioffs[k + 1])]
        if stage_filter and ref.num_inliers < opts.min_num_inliers:
            assert got.config == 0 and got.num_inliers == 0 and len(got_inl) == 0, tuple(pairs[k])
            continue
        tvg_equal(got, ref, tuple(pairs[k]))
        assert (got_inl == ref_inl).all(), tuple(pairs[k])
        configs[ref.config] = configs.get(ref.config, 0) + 1
        n_ok += 1
    return ref_matches, refs, configs, n_ok


def test_config2_shape_calibrated_and_uncalibrated(dsm, oracle):
    """BASELINE configs[1] shape: 4 096-feature images of the bench's own generator (seed 0 = the bench scene), 28
    pairs, ~256 matches and ~165 inliers per pair; E + F + H + pose and the F + H path."""
    n_img = 8
    scene = synthetic.Scene(n_img, 4096, seed=0)
    ims = [scene.image(i) for i in range(n_img)]
    pairs = synthetic.exhaustive_pairs(n_img)
    assert len(pairs) >= 24
    opts = capi.default_two_view_options()
    ref_matches = None
    for prior in (1, 0):
        cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, prior)
                for _ in range(n_img)]
        ref_matches, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, opts, 0, ref_matches)
        assert n_ok >= 24
        assert np.mean([len(x) for x in ref_matches]) > 200
        assert np.mean([r[0].num_inliers for r in refs]) > 120
        assert configs.get(2 if prior else 3, 0) >= 20


@pytest.mark.parametrize("prior", [1, 0])
def test_dense_overlap_more_matches_than_lds_points(dsm, oracle, prior):
    """n_obs = n_feats, pool barely larger: > 2 000 putative matches per pair, so the scoring kernels leave their
    LDS staging (n > 1 536), the local optimisation's tall QR has >= 32 rows per lane, and the ComputeNumTrials
    tables are built for thousands of inlier counts."""
    n_img = 3
    scene = synthetic.Scene(n_img, 4096, seed=77, n_obs=4096, n_pool=4400)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior) for _ in range(n_img)]
    pairs = synthetic.exhaustive_pairs(n_img)
    ref_matches, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, capi.default_two_view_options(), 4)
    assert min(len(x) for x in ref_matches) > 2000
    assert n_ok == len(pairs) and min(r[0].num_inliers for r in refs) > 1200


def test_8192_feature_images_with_verification(dsm, oracle):
    """BASELINE configs[4] image size: 8 192 features per image, calibrated."""
    scene = synthetic.Scene(3, 8192, seed=5)
    ims = [scene.image(i) for i in range(3)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(3)]
    pairs = synthetic.exhaustive_pairs(3)
    ref_matches, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, capi.default_two_view_options(), 8)
    assert n_ok == 3 and min(len(x) for x in ref_matches) > 300


@pytest.mark.parametrize("feats,n_img", [(1024, 4), (4096, 2)])
def test_fixed_4096_trials_per_family(dsm, oracle, feats, n_img):
    """BASELINE configs[4] RANSAC schedule: min_num_trials = max_num_trials = 4 096, confidence
    0.999999, min_inlier_ratio 0.01 so that neither the constructor's cap nor the dynamic stop bites: every family
    runs exactly 4 096 trials."""
    scene = synthetic.Scene(n_img, feats, seed=15)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(n_img)]
    pairs = synthetic.exhaustive_pairs(n_img)
    opts = capi.default_two_view_options(min_num_trials=4096, max_num_trials=4096, confidence=0.999999, min_inlier_ratio=0.01)
    ref_matches, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, opts, 2)
    assert n_ok == len(pairs)
    for ref, _ in refs:
        assert list(ref.num_trials)[:3] == [4096, 4096, 4096], list(ref.num_trials)


def test_planar_and_panoramic_configurations(dsm, oracle):
    """An exact plane seen from different centres ends PLANAR (4), a pure rotation ends PANORAMIC (5)
    (two_view_geometry.cc:266-279: PLANAR_OR_PANORAMIC is split by tvec.norm() == 0)."""
    opts = capi.default_two_view_options()
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(3)]
    pairs = synthetic.exhaustive_pairs(3)
    sp = synthetic.Scene(3, 1024, seed=31, planar=True, planar_depth=0.0)
    ims = [sp.image(i) for i in range(3)]
    _, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, opts, 6)
    assert configs.get(4, 0) >= 2, configs
    so = synthetic.Scene(3, 1024, seed=32, panoramic=True, kp_sigma=0.1)
    ims = [so.image(i) for i in range(3)]
    _, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, opts, 6)
    assert configs.get(5, 0) >= 2, configs


def test_config5_8192_features_with_fixed_4096_trials(dsm, oracle):
    """BASELINE configs[4] as written (missing 2): 8 192 features per image, calibrated E + F + H with
    min_num_trials = max_num_trials = 4 096 per family -- the image size and the RANSAC schedule TOGETHER."""
    scene = synthetic.Scene(3, 8192, seed=21)
    ims = [scene.image(i) for i in range(3)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(3)]
    pairs = synthetic.exhaustive_pairs(3)
    opts = capi.default_two_view_options(min_num_trials=4096, max_num_trials=4096, confidence=0.999999, min_inlier_ratio=0.01)
    ref_matches, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, opts, 12, cache_key="config5")
    assert n_ok == 3 and min(len(x) for x in ref_matches) > 300
    for ref, _ in refs:
        assert list(ref.num_trials)[:3] == [4096, 4096, 4096], list(ref.num_trials)


def test_knn_candidate_pair_list_and_its_shards(dsm, oracle):
    """BASELINE configs[3] pair list (missing 2): a kNN candidate graph -- sparse, images repeated, not
    exhaustive, sorted by query image as VocabTreeFeatureMatcher hands its pairs to Match() (matching.cc:749-839
    runs whatever list it is given) -- through dsm_match_pairs / dsm_verify_pairs against the oracle; then the way
    bench.py --shard-of runs one rank's share: a contiguous slice of the list over ONLY the images it touches,
    re-indexed (bench.py: `remap`), which must give the results of the same pairs under their new ids."""
    n_img = 48
    scene = synthetic.Scene(n_img, 512, seed=9)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(n_img)]
    pairs = synthetic.knn_pairs(scene, n_img, 8, 9)
    assert len(pairs) >= 200 and len(np.unique(pairs)) >= 40
    assert len(pairs) < n_img * (n_img - 1) // 2 // 3                      # sparse
    assert (pairs[:, 0] < pairs[:, 1]).all()
    opts = capi.default_two_view_options()
    ref_matches, refs, configs, n_ok = _compare_stage(dsm, oracle, ims, cams, pairs, opts, 5, cache_key="knn")
    assert n_ok >= 100, (n_ok, configs)
    # one shard of four, the way bench.py builds it
    from dagsfm_amd import sharding
    for r in (1, 3):
        part = sharding.shard(pairs, r, 4)
        used = np.unique(part)
        remap = np.full(n_img, -1, dtype=np.int64)
        remap[used] = np.arange(len(used))
        sub_pairs = remap[part.astype(np.int64)].astype(np.uint32)
        sub_ims = [ims[int(i)] for i in used]
        sub_cams = [cams[int(i)] for i in used]
        lo, hi = sharding.shard_bounds(len(pairs), 4)[r:r + 2]
        _, _, _, n_sub = _compare_stage(dsm, oracle, sub_ims, sub_cams, sub_pairs, opts, 5, ref_matches=ref_matches[lo:hi],
                                        cache_key=("knn-shard", r))
        assert n_sub >= 20
