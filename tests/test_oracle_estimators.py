"""Pins the CPU oracle of the verification path against every known-answer test the reference
holds for it, plus numpy cross-checks of the hand-written Eigen restatements.
Sources of the vectors: /root/reference/src/estimators/{fundamental,essential,homography}_matrix_test.cc,
estimators/utils_test.cc, optim/ransac_test.cc, base/polynomial_test.cc, base/essential_matrix_test.cc."""
import numpy as np

P7_1 = [0.4964, 1.0577, 0.3650, -0.0919, -0.5412, 0.0159, -0.5239, 0.9467, 0.3467, 0.5301, 0.2797, 0.0012, -0.1986, 0.0460]
P7_2 = [0.7570, 2.7340, 0.3961, 0.6981, -0.6014, 0.7110, -0.7385, 2.2712, 0.4177, 1.2132, 0.3052, 0.4835, -0.2171, 0.5057]
P8_1 = [1.839035, 1.924743, 0.543582, 0.375221, 0.473240, 0.142522, 0.964910, 0.598376, 0.102388, 0.140092, 15.994343,
        9.622164, 0.285901, 0.430055, 0.091150, 0.254594]
P8_2 = [1.002114, 1.129644, 1.521742, 1.846002, 1.084332, 0.275134, 0.293328, 0.588992, 0.839509, 0.087290, 1.779735,
        1.116857, 0.878616, 0.602447, 0.642616, 1.028681]
P12_1 = P7_1 + [-0.1622, 0.5347, 0.0796, 0.2379, -0.3946, 0.7969, 0.2, 0.7, 0.6, 0.3]
P12_2 = P7_2 + [-0.2059, 1.1583, 0.0946, 0.7013, -0.6236, 3.0253, 0.5, 0.9, 0.9, 0.2]


def test_seven_point_matlab(oracle):
    # fundamental_matrix_test.cc:39-72 (BOOST_CHECK_CLOSE 1e-6 percent)
    F = oracle.estimate_model(0, P7_1, P7_2)[0]
    ref = np.array([[4.81441976, -8.16978909, 6.73133404], [5.16247992, 0.19325606, -2.87239381],
                    [-9.92570126, 3.64159554, 1.0]])
    assert np.allclose(F, ref, rtol=1e-8, atol=0)


def test_eight_point_matlab(oracle):
    # fundamental_matrix_test.cc:74-105, essential_matrix_test.cc:93-124 (abs 1e-5)
    F = oracle.estimate_model(1, P8_1, P8_2)[0]
    refF = np.array([[-0.217859, 0.419282, -0.0343075], [-0.0717941, 0.0451643, 0.0216073], [0.248062, -0.429478, 0.0221019]])
    assert np.abs(F - refF).max() < 1e-5
    E = oracle.estimate_model(2, P8_1, P8_2)[0]
    refE = np.array([[-0.0811666, 0.255449, -0.0478999], [-0.192392, -0.0531675, 0.119547], [0.177784, -0.22008, -0.015203]])
    assert np.abs(E - refE).max() < 1e-5


def test_five_point_ransac_mask(oracle):
    # essential_matrix_test.cc:47-91: first 10 correspondences inliers, last 2 outliers
    r = oracle.loransac(0, P12_1, P12_2, max_error=0.02, min_inlier_ratio=0.1, confidence=0.9999, min_trials=0,
                        max_trials=2**62, seed=0)
    assert r["success"]
    res = oracle.residuals(0, P12_1, P12_2, r["model"])
    assert (res[:10] <= 0.02 * 0.02).all()
    assert r["mask"][10] == 0 and r["mask"][11] == 0


def test_five_point_models_satisfy_constraints(oracle):
    rng = np.random.default_rng(5)
    for _ in range(20):
        # random relative pose, 5 points in front of both cameras
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0.05, 0.6)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        t = rng.normal(size=3); t /= np.linalg.norm(t)
        X = rng.uniform(-1, 1, (5, 3)) + np.array([0, 0, 5.0])
        x1 = X[:, :2] / X[:, 2:]
        Xc = X @ R.T + t
        x2 = Xc[:, :2] / Xc[:, 2:]
        models = oracle.estimate_model(4, x1, x2)
        assert 1 <= len(models) <= 10
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Etrue = tx @ R
        Etrue /= np.linalg.norm(Etrue)
        best = 1e9
        for E in models:
            assert abs(np.linalg.norm(E) - 1) < 1e-12
            assert abs(np.linalg.det(E)) < 1e-9
            assert np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max() < 1e-8
            h1 = np.c_[x1, np.ones(5)]; h2 = np.c_[x2, np.ones(5)]
            assert np.abs(np.einsum('ij,jk,ik->i', h2, E, h1)).max() < 1e-9
            best = min(best, np.linalg.norm(E - Etrue), np.linalg.norm(E + Etrue))
        assert best < 1e-7


def test_homography_estimate(oracle):
    # homography_matrix_test.cc:41-70
    for x in range(10):
        H0 = np.array([[x, 0.2, 0.3], [30, 0.2, 0.1], [0.3, 20, 1.0]])
        src = np.array([[x, 0], [1, 0], [2, 1], [10, 30]], dtype=np.float64)
        d = np.c_[src, np.ones(4)] @ H0.T
        dst = d[:, :2] / d[:, 2:]
        H = oracle.estimate_model(3, src, dst)[0]
        assert (oracle.residuals(1, src, dst, H) < 1e-6).all()


def test_center_and_normalize_exact(oracle):
    # utils_test.cc:40-61, BOOST_CHECK_EQUAL (exact)
    pts = np.array([[i, i] for i in range(11)], dtype=np.float64)
    normed, M = oracle.center_and_normalize(pts)
    assert M[0, 0] == 0.31622776601683794 and M[1, 1] == 0.31622776601683794
    assert M[0, 2] == -1.5811388300841898 and M[1, 2] == -1.5811388300841898
    assert np.abs(normed.sum(axis=0)).max() < 1e-6


def test_sampson_exact(oracle):
    # utils_test.cc:63-83: E = [t]x R with R = I, t = (1,0,0); residuals exactly 0, 0.5, 2
    E = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)
    r = oracle.residuals(0, [[0, 0]] * 3, [[2, 0], [2, 1], [2, 2]], E)
    assert r.tolist() == [0.0, 0.5, 2.0]


def test_compute_num_trials(oracle):
    # ransac_test.cc:64-83 (SimilarityTransformEstimator<3>: kMinNumSamples = 3)
    assert oracle.compute_num_trials(1, 100, 0.99, 3) == 4605168
    assert oracle.compute_num_trials(10, 100, 0.99, 3) == 4603
    assert oracle.compute_num_trials(10, 100, 0.999, 3) == 6905
    assert oracle.compute_num_trials(100, 100, 0.99, 3) == 1
    assert oracle.compute_num_trials(100, 100, 0.999, 3) == 1
    assert oracle.compute_num_trials(100, 100, 0, 3) == 1
    # caps implied by the default options: E 7071, F 10000 (formula 113174), H 1765, T 6
    assert oracle.compute_num_trials(25000, 100000, 0.999, 5) == 7071
    assert oracle.compute_num_trials(25000, 100000, 0.999, 7) == 113174
    assert oracle.compute_num_trials(25000, 100000, 0.999, 4) == 1765
    assert oracle.compute_num_trials(70000, 100000, 0.999, 1) == 6


def test_polynomial_roots(oracle):
    # polynomial_test.cc:142-190
    re, im = oracle.poly_roots([10, -5, 3, -3, 1])
    assert np.allclose(re, [-0.201826, -0.201826, 0.451826, 0.451826], rtol=1e-5)
    assert np.allclose(im, [0.627696, -0.627696, 0.160867, -0.160867], rtol=1e-5)
    re, im = oracle.poly_roots([10, -5, 3, -3, 0])
    assert np.allclose(re, [0.692438, -0.0962191, -0.0962191, 0], rtol=1e-5, atol=1e-12)
    assert np.allclose(im, [0, 0.651148, -0.651148, 0], rtol=1e-5, atol=1e-12)
    re, im = oracle.poly_roots([1, 2])
    assert re.tolist() == [-2.0]
    re, im = oracle.poly_roots([0, 0, 1, 2])
    assert re.tolist() == [-2.0]
    re, im = oracle.poly_roots([0, 0, 1, 2, 3])
    assert np.allclose(re, [-1, -1]) and np.allclose(np.abs(im), [np.sqrt(2), np.sqrt(2)])
    assert oracle.poly_roots([0, 0, 0])[0] is None


def test_jacobi_svd_against_numpy(oracle):
    rng = np.random.default_rng(1)
    for shape in [(7, 9), (8, 9), (5, 9), (6, 9), (9, 9), (20, 9), (300, 9), (3, 3), (4, 4)]:
        for _ in range(5):
            A = rng.normal(size=shape) * 10 ** rng.uniform(-3, 3)
            U, S, V = oracle.jacobi_svd(A)
            s_np = np.linalg.svd(A, compute_uv=False)
            assert np.allclose(S, s_np, rtol=1e-12, atol=1e-12 * s_np[0])
            assert np.allclose(U @ U.T, np.eye(shape[0]), atol=1e-12)
            assert np.allclose(V @ V.T, np.eye(shape[1]), atol=1e-12)
            k = min(shape)
            assert np.allclose(U[:, :k] * S @ V[:, :k].T, A, rtol=0, atol=1e-11 * s_np[0])
            if shape[1] > shape[0]:  # null space columns
                assert np.abs(A @ V[:, shape[0]:]).max() < 1e-11 * s_np[0]


def test_eigenvalues_against_numpy(oracle):
    rng = np.random.default_rng(2)
    for n in [1, 2, 3, 4, 10]:
        for _ in range(20):
            c = rng.normal(size=n + 1)
            C = np.zeros((n, n))
            for i in range(1, n):
                C[i, i - 1] = 1
            C[0] = -c[1:] / c[0]
            ev = oracle.eigenvalues(C)
            ref = np.linalg.eigvals(C)
            assert np.allclose(np.sort_complex(ev), np.sort_complex(ref), rtol=1e-8, atol=1e-8)
    for _ in range(10):  # general (non-companion) matrices exercise the Hessenberg reduction
        A = rng.normal(size=(6, 6))
        assert np.allclose(np.sort_complex(oracle.eigenvalues(A)), np.sort_complex(np.linalg.eigvals(A)), atol=1e-9)


def _euler(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _cross(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def test_decompose_essential(oracle):
    # base/essential_matrix_test.cc:43-55
    R = _euler(0, 1, 1)
    t = np.array([0.5, 1, 1]); t /= np.linalg.norm(t)
    R1, R2, tt = oracle.decompose_essential(_cross(t) @ R)
    assert min(np.linalg.norm(R1 - R), np.linalg.norm(R2 - R)) < 1e-10
    assert min(np.linalg.norm(tt - t), np.linalg.norm(tt + t)) < 1e-10


def test_pose_from_essential(oracle):
    # base/essential_matrix_test.cc:83-114
    R, t = np.eye(3), np.array([1.0, 0, 0])
    X = np.array([[0, 0, 1], [0, 0.1, 1], [0.1, 0, 1], [0.1, 0.1, 1.0]])
    p1 = X[:, :2] / X[:, 2:]
    Xc = X @ R.T + t
    p2 = Xc[:, :2] / Xc[:, 2:]
    RR, tt, n = oracle.pose_from_essential(_cross(t) @ R, p1, p2)
    assert n == 4 and np.allclose(RR, R) and np.allclose(tt, t)


def test_decompose_homography(oracle):
    # base/homography_matrix_test.cc TestDecomposeHomographyMatrix: H = K2 (R - t n^T / d) K1^-1
    rng = np.random.default_rng(9)
    for _ in range(10):
        R = _euler(*rng.uniform(-0.3, 0.3, 3))
        t = rng.normal(size=3)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        d = rng.uniform(2, 5)
        K = np.array([[640, 0, 320], [0, 640, 240], [0, 0, 1.0]])
        H = K @ (R - np.outer(t, n) / d) @ np.linalg.inv(K)
        Rs, ts, ns = oracle.decompose_homography(H, K, K)
        assert len(Rs) == 4
        err = [np.linalg.norm(Rs[i] - R) + np.linalg.norm(ts[i] * d * (1 if ns[i] @ n > 0 else 1) - t * np.sign(1)) for i in range(4)]
        ok = any(np.linalg.norm(Rs[i] - R) < 1e-6 and np.linalg.norm(np.cross(ns[i], n)) < 1e-6 for i in range(4))
        assert ok, err
    # pure rotation
    Rs, ts, ns = oracle.decompose_homography(K @ _euler(0.1, 0.2, 0.3) @ np.linalg.inv(K), K, K)
    assert len(Rs) == 1 and np.allclose(ts[0], 0) and np.allclose(Rs[0], _euler(0.1, 0.2, 0.3), atol=1e-9)


def test_triangulate_and_quaternion(oracle):
    # base/triangulation_test.cc:42-78
    rng = np.random.default_rng(3)
    R = _euler(0.1, -0.2, 0.3)
    t = np.array([1.0, 0.2, -0.1])
    P1 = np.c_[np.eye(3), np.zeros(3)]
    P2 = np.c_[R, t]
    for _ in range(20):
        X = rng.uniform(-1, 1, 3) + [0, 0, 6]
        x1 = X[:2] / X[2]
        xc = R @ X + t
        x2 = xc[:2] / xc[2]
        assert np.allclose(oracle.triangulate_point(P1, P2, x1, x2), X, atol=1e-9)
    for _ in range(50):
        Rr = _euler(*rng.uniform(-3.1, 3.1, 3))
        q = oracle.rotation_to_quaternion(Rr)
        w, x, y, z = q
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert abs(np.linalg.norm(q) - 1) < 1e-12 and np.allclose(Rq, Rr, atol=1e-12)


def test_image_to_world(oracle):
    from dagsfm_amd import capi
    cam = capi.Camera(model_id=0, has_prior_focal_length=1, width=1000, height=750)
    cam.params[0], cam.params[1], cam.params[2] = 800.0, 500.0, 375.0
    assert oracle.image_to_world(cam, [900.0, 175.0]).tolist() == [0.5, -0.25]
    cam = capi.Camera(model_id=2, has_prior_focal_length=1, width=1000, height=750)
    cam.params[0], cam.params[1], cam.params[2], cam.params[3] = 800.0, 500.0, 375.0, 0.1
    w = oracle.image_to_world(cam, [900.0, 175.0])
    r2 = w @ w
    d = w * (1 + 0.1 * r2)  # distort back
    assert np.allclose(d, [0.5, -0.25], atol=1e-10)


def test_sample_sequence_is_partial_fisher_yates(oracle):
    # random_sampler.cc:50-61 + random.h:122-129: persistent index array, k swaps per draw
    s = oracle.sample_sequence(7, 7, 50, 200)
    assert s.shape == (200, 7) and s.max() < 50
    for row in s:
        assert len(set(row.tolist())) == 7
    assert (oracle.sample_sequence(7, 7, 50, 200) == s).all()
    assert not (oracle.sample_sequence(8, 7, 50, 200) == s).all()


def test_estimate_multiple_two_motions(oracle):
    """EstimateMultiple (two_view_geometry.cc:128-167) on the correspondences of two independently moving
    structures: MULTIPLE, the inlier matches are those of the single passes one after the other (disjoint, and
    each group a subset of the matches); with one structure it degenerates to Estimate plus a DEGENERATE pass."""
    from dagsfm_amd import capi, synthetic

    def pair(seed):
        sc = synthetic.Scene(2, 640, seed=seed, n_pool=900)
        a, b = sc.image(0), sc.image(1)
        return a[1].astype(np.float64), b[1].astype(np.float64), oracle.match_sift_features_cpu(a[0], b[0])

    a1, a2, ma = pair(101)
    b1, b2, mb = pair(202)
    p1, p2 = np.concatenate([a1, b1]), np.concatenate([a2, b2])
    m = np.concatenate([ma, mb + np.array([len(a1), len(a2)], dtype=np.uint32)]).astype(np.uint32)
    cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1)
    opts = capi.default_two_view_options()
    single, inl_single = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, 4)
    opts.multiple_models = 1
    multi, inl = oracle.estimate_two_view_geometry(cam, p1, cam, p2, m, opts, 4)
    assert multi.config == 8 and single.config in (2, 3, 4, 5, 6)
    assert multi.num_inliers > single.num_inliers
    # the first group is exactly the single-pass result (same stream position at the start)
    assert (inl[:single.num_inliers] == inl_single).all()
    as_set = {tuple(r) for r in inl.tolist()}
    assert len(as_set) == len(inl) and as_set <= {tuple(r) for r in m.tolist()}
    assert np.array(multi.E).any() == False and np.array(multi.qvec).any() == False  # fresh TwoViewGeometry()
    # both structures contribute
    first = inl[:, 0] < len(a1)
    assert first.sum() >= 15 and (~first).sum() >= 15


def test_estimate_multiple_ends_when_a_pass_removes_nothing(oracle):
    """min_num_inliers = 0 and fewer than 7 matches: F fails, H succeeds, the inlier list (taken from F's empty mask) is
    empty -- a pass that is not DEGENERATE and removes no match.  The reference's loop (two_view_geometry.cc:128-167)
    would repeat it; the oracle (and the product, tests/test_verify_gpu.py) record the geometry once and stop."""
    from dagsfm_amd import capi
    rng = np.random.default_rng(5)
    camu = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, False)
    opts = capi.default_two_view_options(min_num_inliers=0, multiple_models=1, multiple_ignore_watermark=0)
    p1 = rng.uniform(100, 900, (5, 2))
    p2 = p1 * 1.01 + np.array([4.0, -2.0])
    m = np.stack([np.arange(5), np.arange(5)], axis=1).astype(np.uint32)
    ref, inl = oracle.estimate_two_view_geometry(camu, p1, camu, p2, m, opts, 3)
    assert ref.config == 6 and ref.num_inliers == 0 and len(inl) == 0
    assert ref.num_trials[2] > 0 and ref.num_trials[1] == 0


def _misc_ref():
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmisc_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libmisc_ref.so is built only where /root/reference exists (make -C oracle ref)")
    return ctypes.CDLL(path)


def test_support_measurer_equals_the_references_own_file(oracle):
    """InlierSupportMeasurer::Evaluate / Compare: the oracle's restatement against /root/reference/src/optim/
    support_measurement.cc itself (plain standard C++, compiled where it lies into oracle/_ref/libmisc_ref.so) -- counts and
    residual sums bit for bit on random residual vectors (empty, all in, all out, values at the threshold), and every
    Compare outcome including equal counts with equal / different sums."""
    import ctypes
    R = _misc_ref()
    L = oracle.lib
    for lib in (R, L):
        pre = "ref_" if lib is R else "oracle_"
        getattr(lib, pre + "inlier_support").argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        getattr(lib, pre + "inlier_support_compare").restype = ctypes.c_int
        getattr(lib, pre + "inlier_support_compare").argtypes = [ctypes.c_uint64, ctypes.c_double, ctypes.c_uint64, ctypes.c_double]
    rng = np.random.default_rng(0)
    supports = []
    for trial in range(300):
        n = int(rng.choice([0, 1, 7, 100, 1000]))
        r = np.ascontiguousarray(rng.exponential(4.0, n) ** rng.choice([1, 2]))
        thr = float(rng.choice([0.0, 1.0, 4.0, 16.0, 1e9]))
        if n and trial % 5 == 0:
            r[rng.integers(n)] = thr  # exactly at the threshold: an inlier (<=)
        out = []
        for lib, pre in ((R, "ref_"), (L, "oracle_")):
            cnt, s = ctypes.c_uint64(0), ctypes.c_double(0)
            getattr(lib, pre + "inlier_support")(r.ctypes.data, n, thr, ctypes.byref(cnt), ctypes.byref(s))
            out.append((cnt.value, s.value))
        assert out[0] == out[1], (trial, out)
        supports.append(out[0])
    supports += [(5, 1.0), (5, 1.0), (5, 2.0), (6, 9.0), (0, np.finfo(np.float64).max)]
    for a in supports[::7] + supports[-5:]:
        for b in supports[::11] + supports[-5:]:
            assert R.ref_inlier_support_compare(a[0], a[1], b[0], b[1]) == L.oracle_inlier_support_compare(a[0], a[1], b[0], b[1])
