"""GPU parity of the HIP matcher (through the C-ABI) against the CPU oracle: bit-exact
FeatureMatches.  Patterns follow /root/reference/src/feature/sift_test.cc:300-325, 448-578
(CPU == GPU index-by-index equality)."""
import numpy as np
import pytest

from dagsfm_amd import capi

pytestmark = pytest.mark.gpu


def check_equal(m_gpu, m_cpu):
    # CheckEqualMatches, sift_test.cc:255-262
    assert m_gpu.shape == m_cpu.shape, (m_gpu.shape, m_cpu.shape)
    assert (m_gpu == m_cpu).all()


def rand_sift(rng, n, dup=0):
    """SIFT-like descriptors: 128 x U(0,1)^2, L2-normalised, x512, rounded, saturated to u8."""
    x = rng.random((n, 128), dtype=np.float32) ** 2
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    d = np.clip(np.rint(512.0 * x), 0, 255).astype(np.uint8)
    for _ in range(dup):
        if n >= 2:
            i, j = rng.integers(0, n, 2)
            d[i] = d[j]
    return d


def test_reference_known_answers(dsm, oracle):
    d1 = oracle.create_random_feature_descriptors(2)
    d2 = d1[::-1].copy()
    assert dsm.match_sift_features(d1, d2).tolist() == [[0, 1], [1, 0]]
    empty = np.zeros((0, 128), np.uint8)
    assert len(dsm.match_sift_features(empty, d2)) == 0
    assert len(dsm.match_sift_features(d1, empty)) == 0
    assert len(dsm.match_sift_features(empty, empty)) == 0

    d1 = oracle.create_random_feature_descriptors(100)
    d2 = d1[::-1].copy()
    m = dsm.match_sift_features(d1, d2)
    assert len(m) == 100
    check_equal(m, oracle.match_sift_features_cpu(d1, d2))

    d2 = d1.copy()
    assert len(dsm.match_sift_features(d1, d2)) == 100
    d2[99] = d2[0]
    d2[0, 0] = np.uint8((int(d2[0, 0]) + 50) & 0xFF)
    d2[0] = oracle.l2_normalize_to_u8(d2[0].astype(np.float32))
    d2[99, 0] = np.uint8((int(d2[99, 0]) + 100) & 0xFF)
    d2[99] = oracle.l2_normalize_to_u8(d2[99].astype(np.float32))
    o = capi.default_match_options(max_ratio=0.4)
    m = dsm.match_sift_features(d1[:99], d2, o)
    assert len(m) == 98
    check_equal(m, oracle.match_sift_features_cpu(d1[:99], d2, max_ratio=0.4))
    o = capi.default_match_options(max_ratio=0.5)
    m = dsm.match_sift_features(d1, d2, o)
    assert len(m) == 99
    check_equal(m, oracle.match_sift_features_cpu(d1, d2, max_ratio=0.5))

    d1 = oracle.create_random_feature_descriptors(100)
    d2 = d1.copy()
    d1[0] = d1[1]
    assert len(dsm.match_sift_features(d1, d2, capi.default_match_options(cross_check=0))) == 100
    assert len(dsm.match_sift_features(d1, d2, capi.default_match_options(cross_check=1))) == 98


@pytest.mark.parametrize("n1,n2", [(1, 1), (1, 5), (7, 3), (63, 65), (100, 100), (255, 257), (256, 256),
                                   (300, 1000), (1024, 1024), (1000, 333), (2049, 640)])
@pytest.mark.parametrize("cross", [1, 0])
def test_random_sift_ragged(dsm, oracle, n1, n2, cross):
    rng = np.random.default_rng(1000 * n1 + n2 + cross)
    base = rand_sift(rng, max(n1, n2), dup=3)
    # image 2 = noisy permutation of image 1 plus clutter, so that many matches survive
    d1 = base[:n1]
    perm = rng.permutation(max(n1, n2))[:n2]
    noise = rng.integers(-6, 7, (n2, 128))
    d2 = np.clip(base[perm].astype(np.int32) + noise, 0, 255).astype(np.uint8)
    for ratio, dist in [(0.8, 0.7), (0.95, 1.2), (0.5, 0.3)]:
        o = capi.default_match_options(max_ratio=ratio, max_distance=dist, cross_check=cross)
        m = dsm.match_sift_features(d1, d2, o)
        check_equal(m, oracle.match_sift_features_cpu(d1, d2, ratio, dist, bool(cross)))


def test_full_u8_range_and_ties(dsm, oracle):
    """Arbitrary bytes (dots far above 2^18, saturating acos), exact duplicates, all-zero rows."""
    rng = np.random.default_rng(7)
    d1 = rng.integers(0, 256, (300, 128), dtype=np.uint8)
    d2 = rng.integers(0, 256, (280, 128), dtype=np.uint8)
    d1[5] = 255
    d2[9] = 255
    d2[10] = 255          # duplicate of the best column -> ratio test must reject (sift.cc:151-155)
    d1[17] = 0            # all dots 0 -> best_i2 stays -1 (sift.cc:136)
    d2[33] = 0
    d2[100] = d2[50]
    d1[200] = d2[50]
    for cross in (0, 1):
        for ratio, dist in [(0.8, 0.7), (1.5, 2.0), (0.99, 1.6)]:
            o = capi.default_match_options(max_ratio=ratio, max_distance=dist, cross_check=cross)
            check_equal(dsm.match_sift_features(d1, d2, o),
                        oracle.match_sift_features_cpu(d1, d2, ratio, dist, bool(cross)))
    # low-magnitude descriptors: everything fails max_distance
    d1s = (d1 // 16).astype(np.uint8)
    d2s = (d2 // 16).astype(np.uint8)
    check_equal(dsm.match_sift_features(d1s, d2s), oracle.match_sift_features_cpu(d1s, d2s))
    o = capi.default_match_options(max_ratio=1.0, max_distance=1.7, cross_check=0)
    check_equal(dsm.match_sift_features(d1s, d2s, o), oracle.match_sift_features_cpu(d1s, d2s, 1.0, 1.7, False))


def test_equal_dots_resolve_to_lowest_index(dsm, oracle):
    """Several identical columns in different 32-column tiles and lanes (H2 tie rules)."""
    rng = np.random.default_rng(11)
    d1 = rand_sift(rng, 96)
    d2 = rand_sift(rng, 200)
    for c in (3, 35, 64, 131, 199):
        d2[c] = d1[10]
    d2[77] = d1[20]
    d2[78] = d1[20]
    o = capi.default_match_options(max_ratio=2.0, max_distance=3.0, cross_check=0)  # ratio never rejects
    m = dsm.match_sift_features(d1, d2, o)
    check_equal(m, oracle.match_sift_features_cpu(d1, d2, 2.0, 3.0, False))


def test_pair_list_many_images(dsm, oracle):
    """dsm_set_images + dsm_match_pairs over a ragged image set, all pairs, both orders."""
    rng = np.random.default_rng(3)
    sizes = [0, 1, 40, 256, 300, 513, 1024, 77]
    base = rand_sift(rng, 1100)
    descs = []
    for n in sizes:
        idx = rng.permutation(1100)[:n]
        noise = rng.integers(-5, 6, (n, 128))
        descs.append(np.clip(base[idx].astype(np.int32) + noise, 0, 255).astype(np.uint8))
    dsm.set_images(descs)
    pairs = [(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j]
    pairs.append((2, 2))
    dsm.match_pairs(np.array(pairs, dtype=np.uint32))
    offs, m = dsm.matches()
    counts = dsm.match_counts()
    assert (np.diff(offs.astype(np.int64)) == counts.astype(np.int64)).all()
    for k, (i, j) in enumerate(pairs):
        ref = oracle.match_sift_features_cpu(descs[i], descs[j])
        check_equal(m[int(offs[k]):int(offs[k + 1])], ref)
    ms, nl = dsm.match_kernel_time()
    assert ms > 0 and nl >= 1


def test_4096_features_bit_exact_and_properties(dsm, oracle):
    """BASELINE size (4 096 feats/image): bit-exact on two pairs, plus size-independent
    properties on a block of pairs: ascending idx1, unique idx2, symmetry under image swap."""
    rng = np.random.default_rng(4096)
    pool = rand_sift(rng, 6000)
    descs = []
    for _ in range(4):
        idx = rng.permutation(6000)[:4096]
        noise = rng.integers(-4, 5, (4096, 128))
        descs.append(np.clip(pool[idx].astype(np.int32) + noise, 0, 255).astype(np.uint8))
    dsm.set_images(descs)
    pairs = [(0, 1), (1, 0), (0, 2), (2, 0), (1, 3), (3, 1), (2, 3)]
    dsm.match_pairs(np.array(pairs, dtype=np.uint32))
    offs, m = dsm.matches()
    get = lambda k: m[int(offs[k]):int(offs[k + 1])]
    for k in (0, 6):
        check_equal(get(k), oracle.match_sift_features_cpu(descs[pairs[k][0]], descs[pairs[k][1]]))
    for k in range(len(pairs)):
        mk = get(k)
        assert len(mk) > 1000
        assert (np.diff(mk[:, 0].astype(np.int64)) > 0).all()
        assert len(np.unique(mk[:, 1])) == len(mk)
    for k in (0, 2, 4):  # cross-checked matching is symmetric: swap images <=> swap columns
        a = get(k)
        b = get(k + 1)[:, ::-1]
        b = b[np.argsort(b[:, 0], kind="stable")]
        check_equal(a, b)


def test_more_than_one_column_segment(dsm, oracle):
    """> 4 096 and > 8 192 features: K1 sweeps the columns in segments of 128 tiles and merges them in order;
    ties across segments must still resolve to the lowest index."""
    rng = np.random.default_rng(99)
    d1 = rand_sift(rng, 4500)
    d2 = rand_sift(rng, 9000)
    perm = rng.permutation(4500)
    d2[perm[:3000] * 2] = np.clip(d1[perm[:3000]].astype(np.int32) + rng.integers(-4, 5, (3000, 128)), 0, 255).astype(np.uint8)
    d2[8999] = d2[10]     # duplicate columns in different segments
    d2[4100] = d2[4097]
    d1[7] = d2[10]
    for cross in (1, 0):
        o = capi.default_match_options(cross_check=cross)
        check_equal(dsm.match_sift_features(d1, d2, o), oracle.match_sift_features_cpu(d1, d2, 0.8, 0.7, bool(cross)))
        check_equal(dsm.match_sift_features(d2, d1, o), oracle.match_sift_features_cpu(d2, d1, 0.8, 0.7, bool(cross)))
    o = capi.default_match_options(max_ratio=2.0, max_distance=3.0, cross_check=0)
    check_equal(dsm.match_sift_features(d1, d2, o), oracle.match_sift_features_cpu(d1, d2, 2.0, 3.0, False))


def test_dot4_variant_matches_mfma_kernel(dsm, oracle, monkeypatch):
    """DSM_K1_DOT4=1 (the LDS-tiled v_dot4 comparison variant of pass 1, profiles/r02_k1_variants.md) produces the
    same matches as the MFMA kernel and the oracle, ragged sizes included."""
    from dagsfm_amd import synthetic
    scene = synthetic.Scene(4, 700, seed=12, n_pool=1500)
    ims = [scene.image(i) for i in range(4)]
    descs = [ims[0][0], ims[1][0][:513], ims[2][0][:255], ims[3][0]]
    pairs = synthetic.exhaustive_pairs(4)
    dsm.set_images(descs)
    dsm.match_pairs(pairs)
    offs0, m0 = dsm.matches()
    monkeypatch.setenv("DSM_K1_DOT4", "1")  # a check-build kernel: from here on `dsm` is the check library's context
    dsm.set_images(descs)
    dsm.match_pairs(pairs)
    offs1, m1 = dsm.matches()
    assert (offs0 == offs1).all() and (m0 == m1).all() and len(m0) > 100
    ref = oracle.match_sift_features_cpu(descs[0], descs[3])
    k = [tuple(p) for p in pairs].index((0, 3))
    assert (m1[int(offs1[k]):int(offs1[k + 1])] == ref).all()


@pytest.mark.parametrize("cross", [1, 0])
def test_several_chunks_of_the_pair_list(dsm, oracle, cross, monkeypatch):
    """The matcher works through a long pair list in chunks of bounded scratch (6 GiB of rows by default; the
    2 000-image bench runs 6 of them).  Forced here to a few pairs per chunk: same matches as one chunk and as the
    oracle, with an empty image and a pair listed twice in between."""
    from dagsfm_amd import synthetic
    scene = synthetic.Scene(6, 520, seed=13, n_pool=1400)
    ims = [scene.image(i) for i in range(6)]
    descs = [im[0] for im in ims]
    descs[4] = descs[4][:0]  # no features at all
    pairs = np.array([(0, 1), (2, 1), (0, 3), (4, 5), (1, 4), (5, 0), (0, 1), (3, 2), (5, 2)], dtype=np.uint32)
    o = capi.default_match_options(cross_check=cross)
    dsm.set_images(descs)
    dsm.match_pairs(pairs, o)
    offs0, m0 = dsm.matches()
    monkeypatch.setenv("DSM_MATCH_CHUNK_ROWS", "1400")  # two or three pairs per chunk
    dsm.match_pairs(pairs, o)
    offs1, m1 = dsm.matches()
    assert (offs0 == offs1).all() and (m0 == m1).all()
    for k, (i, j) in enumerate(pairs):
        ref = oracle.match_sift_features_cpu(descs[i], descs[j], cross_check=bool(cross)) if len(descs[i]) and len(descs[j]) \
            else np.zeros((0, 2), np.uint32)
        assert (m1[int(offs1[k]):int(offs1[k + 1])] == ref).all(), (k, i, j)
    assert int(offs1[-1]) > 300


def test_append_images_equals_one_upload(oracle):
    """dsm_append_images: images added behind the resident ones must behave exactly like images uploaded in one
    dsm_set_images call -- same indices, same matches, same verification (the host shim appends what a block of
    ExhaustiveFeatureMatcher's pair list adds instead of uploading the whole set again)."""
    from dagsfm_amd import synthetic
    n_img = 7
    scene = synthetic.Scene(n_img, 700, seed=19, n_pool=2000)
    ims = [scene.image(i) for i in range(n_img)]
    cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, True) for _ in range(n_img)]
    pairs = synthetic.exhaustive_pairs(n_img)
    opts = capi.default_two_view_options()

    def run(ctx):
        ctx.match_pairs(pairs)
        ctx.verify_pairs(opts, user_seed=3)
        offs, m = ctx.matches()
        ioffs, im = ctx.inlier_matches()
        return np.array(offs), np.array(m), np.array(ioffs), np.array(im), [bytes(t) for t in ctx.two_view_geometries()]
    whole = capi.Context(0)
    whole.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    ref = run(whole)
    parts = capi.Context(0)
    parts.set_images([im[0] for im in ims[:3]], [im[1] for im in ims[:3]], cams[:3])
    parts.match_pairs(synthetic.exhaustive_pairs(3))   # results of an earlier call are simply invalidated
    parts.append_images([im[0] for im in ims[3:4]], [im[1] for im in ims[3:4]], cams[3:4])
    parts.append_images([im[0] for im in ims[4:]], [im[1] for im in ims[4:]], cams[4:])
    got = run(parts)
    for a, b in zip(ref[:4], got[:4]):
        assert a.shape == b.shape and (a == b).all()
    assert ref[4] == got[4]
    # keypoints / cameras must come with the appended images iff the resident ones have them
    with pytest.raises(capi.DsmError):
        parts.append_images([ims[0][0]])


@pytest.mark.parametrize("cross", [0, 1])
def test_second_best_in_the_same_column_set_as_the_best(dsm, oracle, cross):
    """Round 3's K1 keeps the top-2 of the TILE MAXIMA of a lane's 16-column sets: `second` misses the second largest value
    inside the set that holds the best, and K1b adds it before the ratio test.  Rows whose runner-up sits (a) in the very
    set of the best column, (b) in the other half-wave's set of the same 32-column tile, (c) in another tile -- close
    enough to the best that the ratio test REJECTS the row only if the true second best is used; plus exact duplicates
    of the best in the same set (a duplicate is the second best, sift.cc:126-132) and runner-ups at LOWER columns."""
    rng = np.random.default_rng(42)
    n1, n2 = 96, 700
    d1 = rand_sift(rng, n1)
    d2 = rand_sift(rng, n2)

    def near(x, k):   # k bytes nudged by one: a neighbour at a small, growing distance
        y = x.astype(np.int32).copy()
        idx = rng.choice(128, k, replace=False)
        y[idx] += np.where(y[idx] < 200, 1, -1)
        return np.clip(y, 0, 255).astype(np.uint8)
    # column sets of a tile: half h holds columns 8q + 4h + e (q, e = 0..3)
    same_set = lambda c: (c & ~31) + 8 * ((((c & 31) >> 3) + 1) % 4) + (c & 7)          # another q, same half, same e
    other_half = lambda c: c ^ 4
    cases = []
    for r in range(0, 90, 3):
        c0 = int(rng.integers(0, n2 - 64))
        kind = r // 3 % 5
        c1 = {0: same_set(c0), 1: other_half(c0), 2: (c0 + 64) % n2, 3: same_set(c0), 4: other_half(c0)}[kind]
        if kind >= 3 and c1 > c0:
            c0, c1 = c1, c0                                   # runner-up (or duplicate) at the LOWER column
        d2[c0] = near(d1[r], 6)
        d2[c1] = d2[c0] if kind == 3 else near(d1[r], 9)      # kind 3: an exact duplicate of the best
        cases.append((r, c0, c1, kind))
    for ratio, dist in [(0.8, 0.7), (0.95, 0.7), (0.6, 1.0)]:
        o = capi.default_match_options(max_ratio=ratio, max_distance=dist, cross_check=cross)
        ref = oracle.match_sift_features_cpu(d1, d2, ratio, dist, bool(cross))
        check_equal(dsm.match_sift_features(d1, d2, o), ref)
        # and the other direction (the planted columns are rows of image a there)
        check_equal(dsm.match_sift_features(d2, d1, o), oracle.match_sift_features_cpu(d2, d1, ratio, dist, bool(cross)))
    # the construction does what it says: with the default ratio most planted rows are rejected although their best
    # column is an almost exact copy
    ref = oracle.match_sift_features_cpu(d1, d2, 0.8, 0.7, False)
    planted = {r for r, _, _, _ in cases}
    assert len(planted - set(ref[:, 0].tolist())) >= 20
