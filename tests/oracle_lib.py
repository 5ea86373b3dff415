"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker; never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_LIB = None

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
f32p = ctypes.POINTER(ctypes.c_float)
f64p = ctypes.POINTER(ctypes.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_match_sift_features_cpu.restype = ctypes.c_int
        lib.oracle_match_sift_features_cpu.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, u8p,
                                                       ctypes.c_int, u8p, ctypes.c_int, u32p]
        lib.oracle_create_random_feature_descriptors.restype = None
        lib.oracle_create_random_feature_descriptors.argtypes = [ctypes.c_int, u8p]
        lib.oracle_l2_normalize_to_u8.restype = None
        lib.oracle_l2_normalize_to_u8.argtypes = [f32p, u8p]
        self._init_two_view()

    # MatchSiftFeaturesCPU, /root/reference/src/feature/sift.cc:810-822
    def match_sift_features_cpu(self, desc1, desc2, max_ratio=0.8, max_distance=0.7, cross_check=True):
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 128)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 128)
        n1, n2 = d1.shape[0], d2.shape[0]
        out = np.zeros((max(n1, 1), 2), dtype=np.uint32)
        n = self.lib.oracle_match_sift_features_cpu(max_ratio, max_distance, int(bool(cross_check)),
                                                    d1.ctypes.data_as(u8p), n1, d2.ctypes.data_as(u8p), n2,
                                                    out.ctypes.data_as(u32p))
        return out[:n].copy()

    # CreateRandomFeatureDescriptors, /root/reference/src/feature/sift_test.cc:243-253
    def create_random_feature_descriptors(self, n):
        out = np.zeros((n, 128), dtype=np.uint8)
        if n:
            self.lib.oracle_create_random_feature_descriptors(n, out.ctypes.data_as(u8p))
        return out

    # ---------------------------------------------------------------- two-view verification
    def _init_two_view(self):
        from dagsfm_amd import capi
        L = self.lib
        cam_p = ctypes.POINTER(capi.Camera)
        L.oracle_estimate_two_view_geometry.restype = None
        L.oracle_estimate_two_view_geometry.argtypes = [cam_p, f64p, ctypes.c_int, cam_p, f64p, ctypes.c_int, u32p,
                                                        ctypes.c_int, ctypes.POINTER(capi.TwoViewOptions),
                                                        ctypes.c_uint32, ctypes.POINTER(capi.TwoViewGeometry), u32p]
        L.oracle_compute_num_trials.restype = ctypes.c_uint64
        L.oracle_compute_num_trials.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, ctypes.c_int]
        L.oracle_estimate_model.restype = ctypes.c_int
        L.oracle_estimate_model.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_int, f64p]
        L.oracle_residuals.restype = None
        L.oracle_residuals.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_int, f64p, f64p]
        L.oracle_center_and_normalize.restype = None
        L.oracle_center_and_normalize.argtypes = [f64p, ctypes.c_int, f64p, f64p]
        L.oracle_poly_roots.restype = ctypes.c_int
        L.oracle_poly_roots.argtypes = [f64p, ctypes.c_int, f64p, f64p]
        L.oracle_jacobi_svd.restype = None
        L.oracle_jacobi_svd.argtypes = [f64p, ctypes.c_int, ctypes.c_int, f64p, f64p, f64p]
        L.oracle_eigenvalues.restype = ctypes.c_int
        L.oracle_eigenvalues.argtypes = [f64p, ctypes.c_int, f64p, f64p]
        L.oracle_loransac.restype = ctypes.c_uint64
        L.oracle_loransac.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, f64p]
        L.oracle_decompose_essential.restype = None
        L.oracle_decompose_essential.argtypes = [f64p, f64p, f64p, f64p]
        L.oracle_pose_from_essential.restype = ctypes.c_int
        L.oracle_pose_from_essential.argtypes = [f64p, f64p, f64p, ctypes.c_int, f64p, f64p]
        L.oracle_decompose_homography.restype = ctypes.c_int
        L.oracle_decompose_homography.argtypes = [f64p, f64p, f64p, f64p, f64p, f64p]
        L.oracle_triangulate_point.restype = None
        L.oracle_triangulate_point.argtypes = [f64p, f64p, f64p, f64p, f64p]
        L.oracle_rotation_to_quaternion.restype = None
        L.oracle_rotation_to_quaternion.argtypes = [f64p, f64p]
        L.oracle_image_to_world.restype = None
        L.oracle_image_to_world.argtypes = [cam_p, f64p, f64p]
        L.oracle_sample_sequence.restype = None
        L.oracle_sample_sequence.argtypes = [ctypes.c_uint32] * 4 + [u32p]

    @staticmethod
    def _d(a):
        return np.ascontiguousarray(a, dtype=np.float64)

    # TwoViewGeometry::Estimate, /root/reference/src/estimators/two_view_geometry.cc:113-126
    def estimate_two_view_geometry(self, cam1, pts1, cam2, pts2, matches, options, seed):
        from dagsfm_amd import capi
        p1, p2 = self._d(pts1).reshape(-1, 2), self._d(pts2).reshape(-1, 2)
        m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
        out = capi.TwoViewGeometry()
        inl = np.zeros((max(len(m), 1), 2), dtype=np.uint32)
        self.lib.oracle_estimate_two_view_geometry(ctypes.byref(cam1), p1.ctypes.data_as(f64p), len(p1),
                                                   ctypes.byref(cam2), p2.ctypes.data_as(f64p), len(p2),
                                                   m.ctypes.data_as(u32p), len(m), ctypes.byref(options),
                                                   ctypes.c_uint32(seed), ctypes.byref(out), inl.ctypes.data_as(u32p))
        return out, inl[:out.num_inliers].copy()

    def match_guided_sift_features_cpu(self, kp1, kp2, desc1, desc2, tvg, max_error=4.0, max_ratio=0.8, max_distance=0.7,
                                       cross_check=True):
        """MatchGuidedSiftFeaturesCPU (sift.cc:824-875): None when the configuration has no guided filter."""
        k1 = np.ascontiguousarray(kp1, dtype=np.float32).reshape(-1, 2)
        k2 = np.ascontiguousarray(kp2, dtype=np.float32).reshape(-1, 2)
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 128)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 128)
        mode = 1 if tvg.config in (2, 3) else (2 if tvg.config in (4, 5, 6) else 0)
        M = np.array(tvg.F if mode == 1 else tvg.H, dtype=np.float64)
        out = np.zeros((max(len(d1), 1), 2), dtype=np.uint32)
        f32p = ctypes.POINTER(ctypes.c_float)
        self.lib.oracle_match_guided_sift_features_cpu.restype = ctypes.c_int
        n = self.lib.oracle_match_guided_sift_features_cpu(
            ctypes.c_double(max_ratio), ctypes.c_double(max_distance), ctypes.c_int(1 if cross_check else 0),
            ctypes.c_double(max_error), k1.ctypes.data_as(f32p), k2.ctypes.data_as(f32p),
            d1.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(len(d1)),
            d2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(len(d2)), ctypes.c_int(mode),
            M.ctypes.data_as(f64p), out.ctypes.data_as(u32p))
        return None if n < 0 else out[:n].copy()

    def compute_num_trials(self, num_inliers, num_samples, confidence, min_samples):
        return int(self.lib.oracle_compute_num_trials(num_inliers, num_samples, confidence, min_samples))

    def estimate_model(self, kind, pts1, pts2):
        p1, p2 = self._d(pts1).reshape(-1, 2), self._d(pts2).reshape(-1, 2)
        out = np.zeros((10, 3, 3))
        n = self.lib.oracle_estimate_model(kind, p1.ctypes.data_as(f64p), p2.ctypes.data_as(f64p), len(p1),
                                           out.ctypes.data_as(f64p))
        return out[:n].copy()

    def residuals(self, kind, pts1, pts2, model):
        p1, p2, M = self._d(pts1).reshape(-1, 2), self._d(pts2).reshape(-1, 2), self._d(model).reshape(9)
        out = np.zeros(len(p1))
        self.lib.oracle_residuals(kind, p1.ctypes.data_as(f64p), p2.ctypes.data_as(f64p), len(p1),
                                  M.ctypes.data_as(f64p), out.ctypes.data_as(f64p))
        return out

    def center_and_normalize(self, pts):
        p = self._d(pts).reshape(-1, 2)
        normed, M = np.zeros_like(p), np.zeros((3, 3))
        self.lib.oracle_center_and_normalize(p.ctypes.data_as(f64p), len(p), normed.ctypes.data_as(f64p),
                                             M.ctypes.data_as(f64p))
        return normed, M

    def poly_roots(self, coeffs):
        c = self._d(coeffs)
        re, im = np.zeros(len(c) + 2), np.zeros(len(c) + 2)
        n = self.lib.oracle_poly_roots(c.ctypes.data_as(f64p), len(c), re.ctypes.data_as(f64p), im.ctypes.data_as(f64p))
        return (None, None) if n < 0 else (re[:n].copy(), im[:n].copy())

    def jacobi_svd(self, A):
        A = self._d(A)
        r, c = A.shape
        U, S, V = np.zeros((r, r)), np.zeros(min(r, c)), np.zeros((c, c))
        self.lib.oracle_jacobi_svd(A.ctypes.data_as(f64p), r, c, U.ctypes.data_as(f64p), S.ctypes.data_as(f64p),
                                   V.ctypes.data_as(f64p))
        return U, S, V

    def eigenvalues(self, A):
        A = self._d(A)
        n = A.shape[0]
        re, im = np.zeros(n), np.zeros(n)
        k = self.lib.oracle_eigenvalues(A.ctypes.data_as(f64p), n, re.ctypes.data_as(f64p), im.ctypes.data_as(f64p))
        return None if k < 0 else re + 1j * im

    def loransac(self, family, pts1, pts2, max_error, min_inlier_ratio=0.25, confidence=0.999, min_trials=30,
                 max_trials=10000, seed=0):
        p1, p2 = self._d(pts1).reshape(-1, 2), self._d(pts2).reshape(-1, 2)
        ok, nt = ctypes.c_int(0), ctypes.c_uint64(0)
        mask = ctypes.create_string_buffer(max(len(p1), 1))
        model = np.zeros(9)
        ninl = self.lib.oracle_loransac(family, p1.ctypes.data_as(f64p), p2.ctypes.data_as(f64p), len(p1), max_error,
                                        min_inlier_ratio, confidence, min_trials, max_trials, seed, ctypes.byref(ok),
                                        ctypes.byref(nt), mask, model.ctypes.data_as(f64p))
        m = np.frombuffer(mask.raw, dtype=np.uint8)[:len(p1)].copy() if ok.value else np.zeros(0, np.uint8)
        return dict(success=bool(ok.value), num_trials=nt.value, num_inliers=int(ninl), mask=m, model=model)

    def decompose_essential(self, E):
        E = self._d(E)
        R1, R2, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
        self.lib.oracle_decompose_essential(E.ctypes.data_as(f64p), R1.ctypes.data_as(f64p), R2.ctypes.data_as(f64p),
                                            t.ctypes.data_as(f64p))
        return R1, R2, t

    def pose_from_essential(self, E, pts1, pts2):
        E, p1, p2 = self._d(E), self._d(pts1).reshape(-1, 2), self._d(pts2).reshape(-1, 2)
        R, t = np.zeros((3, 3)), np.zeros(3)
        n = self.lib.oracle_pose_from_essential(E.ctypes.data_as(f64p), p1.ctypes.data_as(f64p), p2.ctypes.data_as(f64p),
                                                len(p1), R.ctypes.data_as(f64p), t.ctypes.data_as(f64p))
        return R, t, n

    def decompose_homography(self, H, K1, K2):
        H, K1, K2 = self._d(H), self._d(K1), self._d(K2)
        R, t, n = np.zeros((4, 3, 3)), np.zeros((4, 3)), np.zeros((4, 3))
        k = self.lib.oracle_decompose_homography(H.ctypes.data_as(f64p), K1.ctypes.data_as(f64p), K2.ctypes.data_as(f64p),
                                                 R.ctypes.data_as(f64p), t.ctypes.data_as(f64p), n.ctypes.data_as(f64p))
        return R[:k], t[:k], n[:k]

    def triangulate_point(self, P1, P2, p1, p2):
        P1, P2, p1, p2 = self._d(P1), self._d(P2), self._d(p1), self._d(p2)
        X = np.zeros(3)
        self.lib.oracle_triangulate_point(P1.ctypes.data_as(f64p), P2.ctypes.data_as(f64p), p1.ctypes.data_as(f64p),
                                          p2.ctypes.data_as(f64p), X.ctypes.data_as(f64p))
        return X

    def rotation_to_quaternion(self, R):
        R = self._d(R)
        q = np.zeros(4)
        self.lib.oracle_rotation_to_quaternion(R.ctypes.data_as(f64p), q.ctypes.data_as(f64p))
        return q

    def image_to_world(self, cam, p):
        p = self._d(p)
        w = np.zeros(2)
        self.lib.oracle_image_to_world(ctypes.byref(cam), p.ctypes.data_as(f64p), w.ctypes.data_as(f64p))
        return w

    def sample_sequence(self, seed, k, total, n_draws):
        out = np.zeros((n_draws, k), dtype=np.uint32)
        self.lib.oracle_sample_sequence(seed, k, total, n_draws, out.ctypes.data_as(u32p))
        return out

    def l2_normalize_to_u8(self, row):
        r = np.ascontiguousarray(row, dtype=np.float32).reshape(128)
        out = np.zeros(128, dtype=np.uint8)
        self.lib.oracle_l2_normalize_to_u8(r.ctypes.data_as(f32p), out.ctypes.data_as(u8p))
        return out


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = Oracle(ctypes.CDLL(path))
    return _LIB


def load_native(build_dir=None):
    """The same sources built -O3 -march=native ON THIS HOST (oracle/Makefile `native`): a second, labelled CPU
    baseline for bench.py; returns None when it cannot be built here.  Never used as the parity checker."""
    import tempfile
    d = build_dir or os.path.join(tempfile.gettempdir(), "dagsfm_oracle_native_%d" % os.getuid())
    try:
        subprocess.check_call(["make", "-s", "-C", _DIR, "native", "NDIR=" + d], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return Oracle(ctypes.CDLL(os.path.join(d, "liboracle_native.so")))
    except Exception:
        return None


class OracleVocabulary(ctypes.Structure):
    _fields_ = [("num_words", ctypes.c_uint32), ("words", ctypes.c_void_p), ("proj", ctypes.c_void_p), ("thresholds", ctypes.c_void_p)]


class RetrievalOracle:
    """oracle/retrieval.cc: VisualIndex Add / Prepare / Query restated (exact nearest words)."""

    def __init__(self, words, projection, thresholds):
        self.L = load().lib
        L = self.L
        L.oracle_retrieval_create.restype = ctypes.c_void_p
        L.oracle_retrieval_create.argtypes = [ctypes.POINTER(OracleVocabulary)]
        L.oracle_retrieval_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_retrieval_find_word_ids.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        L.oracle_retrieval_add.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32]
        L.oracle_retrieval_prepare.argtypes = [ctypes.c_void_p]
        L.oracle_retrieval_query.restype = ctypes.c_uint32
        L.oracle_retrieval_query.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        self._keep = (np.ascontiguousarray(words, np.uint8).reshape(-1, 128), np.ascontiguousarray(projection, np.float32).reshape(64, 128),
                      np.ascontiguousarray(thresholds, np.float32).reshape(-1, 64))
        v = OracleVocabulary(num_words=self._keep[0].shape[0], words=self._keep[0].ctypes.data, proj=self._keep[1].ctypes.data,
                             thresholds=self._keep[2].ctypes.data)
        self.h = L.oracle_retrieval_create(ctypes.byref(v))

    def __del__(self):
        try:
            self.L.oracle_retrieval_destroy(self.h)
        except Exception:
            pass

    def find_word_ids(self, desc, k):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 128)
        out = np.zeros((len(d), k), np.int32)
        self.L.oracle_retrieval_find_word_ids(self.h, d.ctypes.data, len(d), k, out.ctypes.data)
        return out

    def add(self, image_id, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 128)
        self.L.oracle_retrieval_add(self.h, image_id, d.ctypes.data, len(d))

    def prepare(self):
        self.L.oracle_retrieval_prepare(self.h)

    def query(self, desc, num_neighbors=5, max_num_images=-1, capacity=100000):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 128)
        ids = np.zeros(capacity, np.int32)
        sc = np.zeros(capacity, np.float32)
        n = self.L.oracle_retrieval_query(self.h, d.ctypes.data, len(d), num_neighbors, max_num_images, ids.ctypes.data, sc.ctypes.data, capacity)
        return ids[:n].copy(), sc[:n].copy()


def _retrieval_oracle_add_geom(self, image_id, desc, geom):
    """VisualIndex::Add with the features' geometry (x, y, scale, orientation) kept in the entries."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 128)
    g = np.ascontiguousarray(geom, np.float32).reshape(-1, 4)
    assert len(g) == len(d)
    self.L.oracle_retrieval_add_geom.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    self.L.oracle_retrieval_add_geom(self.h, image_id, d.ctypes.data, len(d), g.ctypes.data)


def _retrieval_oracle_query_verified(self, desc, geom, num_neighbors=5, max_num_images=-1, num_images_after_verification=0, capacity=100000):
    """VisualIndex::Query with geometries: retrieval + spatial verification + re-ranking (visual_index.h:259-500)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 128)
    g = np.ascontiguousarray(geom, np.float32).reshape(-1, 4)
    ids = np.zeros(capacity, np.int32)
    sc = np.zeros(capacity, np.float32)
    self.L.oracle_retrieval_query_verified.restype = ctypes.c_uint32
    self.L.oracle_retrieval_query_verified.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    n = self.L.oracle_retrieval_query_verified(self.h, d.ctypes.data, g.ctypes.data, len(d), num_neighbors, max_num_images,
                                               num_images_after_verification, ids.ctypes.data, sc.ctypes.data, capacity)
    return ids[:n].copy(), sc[:n].copy()


def _retrieval_oracle_use_flann(self, flann_index, num_checks=256):
    """Every later Add / Query asks the reference's own FLANN (tests/flann_ref.Index: libflann_ref.so, compiled from
    /root/reference/lib/FLANN) for its word ids instead of the exact search -- a plain C function pointer, no Python in the
    loop.  None restores the exact search."""
    from tests import flann_ref
    self.L.oracle_retrieval_set_word_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    if flann_index is None:
        self.L.oracle_retrieval_set_word_search(self.h, None, None)
        return
    flann_ref.load().flann_ref_set_search(flann_index.h, num_checks, 1)
    fn = ctypes.cast(flann_ref.load().flann_ref_find_word_ids, ctypes.c_void_p)
    self._flann = flann_index
    self.L.oracle_retrieval_set_word_search(self.h, fn, flann_index.h)


RetrievalOracle.use_flann = _retrieval_oracle_use_flann
RetrievalOracle.add_geom = _retrieval_oracle_add_geom
RetrievalOracle.query_verified = _retrieval_oracle_query_verified


def keypoint_geometry(kp):
    """(x, y, FeatureKeypoint::ComputeScale(), ComputeOrientation()) of keypoints with 2 (x, y), 4 (x, y, scale,
    orientation) or 6 (x, y, a11, a12, a21, a22) columns (feature/types.cc:42-98), computed by the oracle in C++ float
    (numpy's float32 arctan2 is not glibc's atan2f to the last bit)."""
    k = np.ascontiguousarray(kp, np.float32)
    if len(k) == 0:
        return np.zeros((0, 4), np.float32)
    k = k.reshape(len(k), -1)
    n = len(k)
    k6 = np.zeros((n, 6), np.float32)
    k6[:, :2] = k[:, :2]
    if k.shape[1] == 2:
        k6[:, 2], k6[:, 5] = 1.0, 1.0
    elif k.shape[1] == 4:  # FeatureKeypoint(x, y, scale, orientation), types.cc:50-58 -- only used with exact inputs in the tests
        s, o = k[:, 2], k[:, 3]
        k6[:, 2], k6[:, 3], k6[:, 4], k6[:, 5] = s * np.cos(o), -s * np.sin(o), s * np.sin(o), s * np.cos(o)
    else:
        k6[:, 2:] = k[:, 2:6]
    out = np.zeros((n, 4), np.float32)
    L = load().lib
    L.oracle_sv_keypoint_geometry.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    L.oracle_sv_keypoint_geometry(k6.ctypes.data, n, out.ctypes.data)
    return out


def sv_transform_from_match(g1, g2):
    L = load().lib
    a, b = np.ascontiguousarray(g1, np.float32), np.ascontiguousarray(g2, np.float32)
    out = np.zeros(4, np.float32)
    L.oracle_sv_transform_from_match.argtypes = [ctypes.c_void_p] * 3
    L.oracle_sv_transform_from_match(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def sv_estimate_affine(x1, x2):
    L = load().lib
    a, b = np.ascontiguousarray(x1, np.float64).reshape(-1, 2), np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
    out = np.zeros(6, np.float64)
    L.oracle_sv_estimate_affine.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    L.oracle_sv_estimate_affine(a.ctypes.data, b.ctypes.data, len(a), out.ctypes.data)
    return out.reshape(2, 3)


def sv_vote_and_verify(g1, g2, platform_order=True):
    """VoteAndVerify of 1-to-1 matches; platform_order=False: equal bin scores by ascending bin index instead of libstdc++'s
    hash-table order (only for measuring how often that matters)."""
    L = load().lib
    a, b = np.ascontiguousarray(g1, np.float32).reshape(-1, 4), np.ascontiguousarray(g2, np.float32).reshape(-1, 4)
    L.oracle_sv_vote_and_verify_order.restype = ctypes.c_int
    L.oracle_sv_vote_and_verify_order.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return int(L.oracle_sv_vote_and_verify_order(len(a), a.ctypes.data, b.ctypes.data, 1 if platform_order else 0))
