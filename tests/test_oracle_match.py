"""Pins the CPU oracle of the matcher against the reference's own known-answer tests
(/root/reference/src/feature/sift_test.cc:300-325 and :505-571)."""
import numpy as np


def test_match_two_reversed(oracle):
    # TestMatchSiftFeaturesCPU, sift_test.cc:300-325
    d1 = oracle.create_random_feature_descriptors(2)
    d2 = d1[::-1].copy()  # colwise().reverse() reverses the row order
    m = oracle.match_sift_features_cpu(d1, d2)
    assert m.tolist() == [[0, 1], [1, 0]]
    empty = oracle.create_random_feature_descriptors(0)
    assert len(oracle.match_sift_features_cpu(empty, d2)) == 0
    assert len(oracle.match_sift_features_cpu(d1, empty)) == 0
    assert len(oracle.match_sift_features_cpu(empty, empty)) == 0


def test_reversed_100(oracle):
    # sift_test.cc:513-522: 100 matches
    d1 = oracle.create_random_feature_descriptors(100)
    d2 = d1[::-1].copy()
    m = oracle.match_sift_features_cpu(d1, d2)
    assert len(m) == 100
    assert (m[:, 0] == np.arange(100)).all() and (m[:, 1] == 99 - np.arange(100)).all()


def test_ratio_test_counts(oracle):
    # sift_test.cc:524-549: 100, then 98 (max_ratio 0.4, top 99 rows), then 99 (max_ratio 0.5)
    d1 = oracle.create_random_feature_descriptors(100)
    d2 = d1.copy()
    assert len(oracle.match_sift_features_cpu(d1, d2)) == 100
    d2[99] = d2[0]
    # descriptors2(0,0) += 50.0f on a uint8 matrix wraps like uint8 arithmetic
    d2[0, 0] = np.uint8((int(d2[0, 0]) + 50) & 0xFF)
    d2[0] = oracle.l2_normalize_to_u8(d2[0].astype(np.float32))
    d2[99, 0] = np.uint8((int(d2[99, 0]) + 100) & 0xFF)
    d2[99] = oracle.l2_normalize_to_u8(d2[99].astype(np.float32))
    assert len(oracle.match_sift_features_cpu(d1[:99], d2, max_ratio=0.4)) == 98
    assert len(oracle.match_sift_features_cpu(d1, d2, max_ratio=0.5)) == 99


def test_cross_check_counts(oracle):
    # sift_test.cc:551-569: 100 without cross check, 98 with
    d1 = oracle.create_random_feature_descriptors(100)
    d2 = d1.copy()
    d1[0] = d1[1]
    assert len(oracle.match_sift_features_cpu(d1, d2, cross_check=False)) == 100
    assert len(oracle.match_sift_features_cpu(d1, d2, cross_check=True)) == 98


def test_second_best_duplicate_rejected(oracle):
    # sift.cc:151-155: best == second best must fail the ratio test (>=)
    d1 = oracle.create_random_feature_descriptors(4)
    d2 = np.concatenate([d1, d1[:1]], axis=0)  # column 4 duplicates column 0
    m = oracle.match_sift_features_cpu(d1, d2, cross_check=False)
    assert 0 not in m[:, 0].tolist()
    assert sorted(m[:, 0].tolist()) == [1, 2, 3]


def test_guided_matching_filters_by_geometry(oracle):
    """MatchGuidedSiftFeaturesCPU (sift.cc:824-875): with the true fundamental matrix of a synthetic pair, keypoint
    pairs off the epipolar geometry get distance 0 -- geometric outliers whose descriptors match disappear and
    no surviving match violates the filter; a configuration without filter leaves the matches alone."""
    from dagsfm_amd import capi, synthetic
    sc = synthetic.Scene(2, 512, seed=5, n_pool=700)
    a, b = sc.image(0), sc.image(1)
    plain = oracle.match_sift_features_cpu(a[0], b[0])
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 0)
    opts = capi.default_two_view_options()
    tv, inl = oracle.estimate_two_view_geometry(cam, a[1].astype(np.float64), cam, b[1].astype(np.float64), plain, opts, 3)
    assert tv.config == 3
    g = oracle.match_guided_sift_features_cpu(a[1], b[1], a[0], b[0], tv)
    assert g is not None and len(g) >= len(inl) - 5 and len(g) <= len(plain)
    F = np.array(tv.F, dtype=np.float32).reshape(3, 3)
    x1 = np.c_[a[1][g[:, 0]], np.ones(len(g), dtype=np.float32)]
    x2 = np.c_[b[1][g[:, 1]], np.ones(len(g), dtype=np.float32)]
    Fx1, Ftx2 = x1 @ F.T, x2 @ F
    samp = (np.sum(x2 * Fx1, axis=1) ** 2) / (Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2 + Ftx2[:, 0] ** 2 + Ftx2[:, 1] ** 2)
    assert (samp <= 16.0 * 1.001).all()
    # every guided match is mutual-best under the mask, so plain matches that satisfy the geometry survive
    plain_set = {tuple(r) for r in plain.tolist()}
    assert len(plain_set & {tuple(r) for r in g.tolist()}) >= len(inl) - 5
    tv.config = 1  # DEGENERATE: no filter (sift.cc:861-863)
    assert oracle.match_guided_sift_features_cpu(a[1], b[1], a[0], b[0], tv) is None
