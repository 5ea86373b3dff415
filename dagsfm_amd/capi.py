"""ctypes binding of the C-ABI in include/dagsfm_mi355x.h (no torch types cross it)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSM_LIB_PATH: developer override (e.g. the -DDSM_PROFILE_SECTIONS build under dagsfm_amd/prof/)
LIB_PATH = os.environ.get("DSM_LIB_PATH") or os.path.join(_HERE, "libdagsfm_mi355x.so")
# the same sources with -DDSM_CHECK_BUILD: the product plus the cross-check schedules (csrc/ctx.h, csrc/Makefile).  Only
# tools/ and the schedule-parametrised tests load it -- Context(check=True), or any check-only DSM_* variable in the process
# environment when the context is created; everything else, bench.py and the shim included, runs the product library.
CHECK_LIB_PATH = os.environ.get("DSM_CHECK_LIB_PATH") or os.path.join(_HERE, "libdagsfm_mi355x_check.so")

# the RCCL companion (include/dagsfm_gather.h): multi-GPU assembly of the match graph below the host language
GATHER_LIB_PATH = os.path.join(_HERE, "libdagsfm_gather.so")

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)
f32p = ctypes.POINTER(ctypes.c_float)


class DsmError(RuntimeError):
    pass


class MatchOptions(ctypes.Structure):
    """SiftMatchingOptions (matching half), /root/reference/src/feature/sift.h:116-165."""
    _fields_ = [("max_ratio", ctypes.c_double), ("max_distance", ctypes.c_double),
                ("cross_check", ctypes.c_int32), ("max_num_matches", ctypes.c_int32)]


class TwoViewOptions(ctypes.Structure):
    """TwoViewGeometry::Options + RANSACOptions, two_view_geometry.h:105-157, ransac.h:47-72."""
    _fields_ = [("min_num_inliers", ctypes.c_uint64), ("min_E_F_inlier_ratio", ctypes.c_double),
                ("max_H_inlier_ratio", ctypes.c_double), ("watermark_min_inlier_ratio", ctypes.c_double),
                ("watermark_border_size", ctypes.c_double), ("detect_watermark", ctypes.c_int32),
                ("multiple_models", ctypes.c_int32), ("max_error", ctypes.c_double),
                ("min_inlier_ratio", ctypes.c_double), ("confidence", ctypes.c_double),
                ("min_num_trials", ctypes.c_uint64), ("max_num_trials", ctypes.c_uint64),
                ("multiple_ignore_watermark", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class Camera(ctypes.Structure):
    _fields_ = [("model_id", ctypes.c_int32), ("has_prior_focal_length", ctypes.c_int32),
                ("width", ctypes.c_uint64), ("height", ctypes.c_uint64), ("params", ctypes.c_double * 12)]


class TwoViewGeometry(ctypes.Structure):
    _fields_ = [("config", ctypes.c_int32), ("num_inliers", ctypes.c_uint32), ("num_matches", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("F", ctypes.c_double * 9), ("E", ctypes.c_double * 9),
                ("H", ctypes.c_double * 9), ("qvec", ctypes.c_double * 4), ("tvec", ctypes.c_double * 3),
                ("tri_angle", ctypes.c_double), ("num_trials", ctypes.c_uint32 * 4),
                ("num_models", ctypes.c_uint32 * 4)]


class DeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("arch", ctypes.c_char * 64), ("compute_units", ctypes.c_int32),
                ("clock_khz", ctypes.c_int32), ("memory_clock_khz", ctypes.c_int32), ("memory_bus_bits", ctypes.c_int32),
                ("total_memory", ctypes.c_uint64), ("l2_bytes", ctypes.c_int32), ("lds_per_cu", ctypes.c_int32)]


class Vocabulary(ctypes.Structure):
    """dsm_vocabulary: visual words, Hamming-embedding projection and per-word thresholds (visual_index.h / inverted_index.h)."""
    _fields_ = [("num_words", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("words", ctypes.c_void_p),
                ("projection", ctypes.c_void_p), ("thresholds", ctypes.c_void_p)]


_libs = {}


def lib(check=False):
    """Loads the shared library (check=True: the check build); raises if it has not been built (no fallback)."""
    if check not in _libs:
        path = CHECK_LIB_PATH if check else LIB_PATH
        if not os.path.exists(path):
            raise DsmError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C dagsfm_amd/csrc)" % path)
        L = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        L.dsm_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.dsm_ctx_destroy.argtypes = [vp]
        L.dsm_ctx_destroy.restype = None
        L.dsm_last_error.argtypes = [vp]
        L.dsm_last_error.restype = ctypes.c_char_p
        L.dsm_sync.argtypes = [vp]
        L.dsm_set_images.argtypes = [vp, ctypes.c_uint32, u32p, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                     ctypes.c_uint32, ctypes.POINTER(Camera)]
        L.dsm_match_pairs.argtypes = [vp, ctypes.c_uint32, u32p, ctypes.POINTER(MatchOptions)]
        L.dsm_get_match_counts.argtypes = [vp, vp]
        L.dsm_get_matches.argtypes = [vp, vp, vp, ctypes.c_uint64]
        L.dsm_match_sift_features.argtypes = [vp, ctypes.POINTER(MatchOptions), u8p, ctypes.c_uint32, u8p,
                                              ctypes.c_uint32, u32p, u32p]
        L.dsm_get_match_kernel_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double), u32p]
        L.dsm_get_match_resolve_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.dsm_pair_seed.argtypes = [ctypes.c_uint32] * 3
        L.dsm_pair_seed.restype = ctypes.c_uint32
        L.dsm_verify_pairs.argtypes = [vp, ctypes.POINTER(TwoViewOptions), u32p, ctypes.c_uint32, ctypes.c_int32]
        L.dsm_guided_match_pairs.argtypes = [vp, ctypes.POINTER(MatchOptions), ctypes.POINTER(TwoViewOptions), ctypes.c_int32]
        L.dsm_get_two_view_geometries.argtypes = [vp, vp]
        L.dsm_get_inlier_matches.argtypes = [vp, vp, vp, ctypes.c_uint64]
        L.dsm_get_verify_kernel_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.dsm_estimate_two_view_geometry.argtypes = [vp, ctypes.POINTER(Camera), ctypes.POINTER(ctypes.c_double),
                                                     ctypes.c_uint32, ctypes.POINTER(Camera),
                                                     ctypes.POINTER(ctypes.c_double), ctypes.c_uint32, u32p, ctypes.c_uint32,
                                                     ctypes.POINTER(TwoViewOptions), ctypes.c_uint32,
                                                     ctypes.POINTER(TwoViewGeometry), u32p]
        L.dsm_debug_sample_sequence.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u32p]
        L.dsm_get_device_info.argtypes = [vp, ctypes.POINTER(DeviceInfo)]
        L.dsm_get_match_gather_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.dsm_get_match_tail_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.dsm_retrieval_set_vocabulary.argtypes = [vp, ctypes.POINTER(Vocabulary)]
        L.dsm_retrieval_index.argtypes = [vp]
        L.dsm_retrieval_query.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp, vp]
        L.dsm_retrieval_debug_word_ids.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp]
        L.dsm_get_retrieval_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.dsm_view_graph_filter_cycles.argtypes = [vp, ctypes.c_uint32, vp, vp, ctypes.c_double, vp, vp]
        L.dsm_debug_image_to_world.argtypes = [vp, ctypes.POINTER(Camera), ctypes.c_uint32, ctypes.POINTER(ctypes.c_double),
                                               ctypes.POINTER(ctypes.c_double)]
        L.dsm_default_match_options.argtypes = [ctypes.POINTER(MatchOptions)]
        L.dsm_default_match_options.restype = None
        L.dsm_default_two_view_options.argtypes = [ctypes.POINTER(TwoViewOptions)]
        L.dsm_default_two_view_options.restype = None
        L.dsm_set_debug_option.restype = ctypes.c_int
        L.dsm_set_debug_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
        _libs[check] = L
    return _libs[check]


def pair_seed(id1, id2, user_seed=0):
    return int(lib().dsm_pair_seed(id1, id2, user_seed))


def simple_pinhole(f, cx, cy, width, height, prior=True):
    c = Camera(model_id=0, has_prior_focal_length=int(bool(prior)), width=width, height=height)
    c.params[0], c.params[1], c.params[2] = f, cx, cy
    return c


CAMERA_MODEL_NUM_PARAMS = (3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12)  # camera_models.h:187-349, ids 0..10


def camera(model_id, params, width, height, prior=True):
    """dsm_camera of any of the reference's camera models (params in the reference's order)."""
    c = Camera(model_id=model_id, has_prior_focal_length=int(bool(prior)), width=width, height=height)
    for k, v in enumerate(params):
        c.params[k] = float(v)
    return c


def default_match_options(**kw):
    o = MatchOptions()
    lib().dsm_default_match_options(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def default_two_view_options(**kw):
    o = TwoViewOptions()
    lib().dsm_default_two_view_options(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# the keys dsm_set_debug_option knows (csrc/ctx.h): scheduling knobs of the product, and the cross-check switches of the check build
PRODUCT_OPTION_KEYS = ("DSM_MATCH_CHUNK_ROWS", "DSM_VERIFY_CHUNK_PAIRS", "DSM_VERIFY_LANES", "DSM_VERIFY_INLINE_LO", "DSM_VERIFY_ITEM_MODE",
                       "DSM_LO_TAIL", "DSM_LO_TAIL_MODE", "DSM_VERIFY_GRID_DIV")
CHECK_OPTION_KEYS = ("DSM_K1_DOT4", "DSM_VERIFY_DEBUG", "DSM_SAMPLER_SERIAL", "DSM_LO_PREPARE_WAVE", "DSM_LO_JACOBI_GROUPS", "DSM_ROOTS_LDS",
                     "DSM_FINAL_WAVES", "DSM_VERIFY_LEGACY", "DSM_VERIFY_FIXED_BATCH", "DSM_VERIFY_LANE_SPLIT", "DSM_DEBUG_SAMPLER_MODE",
                     "DSM_VOCAB_ASSIGN_VALU", "DSM_VERIFY_HOST_LOOP", "DSM_SCORE_PREFILTER", "DSM_VERIFY_REPLAY_GRID", "DSM_REPLAY_LEGACY", "DSM_FLANN_GROUP", "DSM_FLANN_STATS", "DSM_ELU_LDS", "DSM_HYP_GRID", "DSM_SPEC_MARGIN")
DEBUG_OPTION_KEYS = PRODUCT_OPTION_KEYS  # what a deployer's library knows


def check_requested():
    """True when the process environment carries a cross-check switch (or DSM_LIBRARY=check): contexts created now use the check build."""
    return os.environ.get("DSM_LIBRARY") == "check" or any(os.environ.get(k) is not None for k in CHECK_OPTION_KEYS)


class Context:
    """One context = one GPU (SiftFeatureMatcher + FeatureMatcherCache of the reference)."""

    def __init__(self, device=0, check=None):
        self._handle = ctypes.c_void_p()
        self._debug = {}
        self.check = check_requested() if check is None else bool(check)
        self._L = lib(self.check)
        rc = self._L.dsm_ctx_create(device, ctypes.byref(self._handle))
        if rc != 0:
            raise DsmError("dsm_ctx_create failed (%d): %s" % (rc, self._L.dsm_last_error(None).decode()))
        self.n_pairs = 0

    @property
    def _h(self):
        """The context handle.  The library itself never reads the environment (dsm_set_debug_option is the only way in);
        this TEST / TOOL binding forwards the DSM_* debug variables of the process to the context before every call, so
        that `DSM_VERIFY_LANES=1 python tools/...` and monkeypatch.setenv in the tests keep working."""
        if self._handle:
            for key in PRODUCT_OPTION_KEYS + CHECK_OPTION_KEYS:  # a check-only switch on a product context fails loudly
                want = os.environ.get(key)
                if self._debug.get(key) != want:
                    self.set_debug_option(key, want)
        return self._handle

    def set_memory_budget(self, nbytes):
        """dsm_ctx_set_memory_budget: bytes of transient chunk scratch the matcher and the verifier may hold (0: the defaults)."""
        self._L.dsm_ctx_set_memory_budget.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        self._chk(self._L.dsm_ctx_set_memory_budget(self._handle, ctypes.c_uint64(int(nbytes))))

    def memory_footprint(self):
        """dsm_ctx_memory_footprint -> (resident_bytes, scratch_bytes) the context holds on the device right now."""
        r, s = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._L.dsm_ctx_memory_footprint.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        self._chk(self._L.dsm_ctx_memory_footprint(self._handle, ctypes.byref(r), ctypes.byref(s)))
        return int(r.value), int(s.value)

    def set_debug_option(self, key, value):
        """dsm_set_debug_option: a scheduling / cross-check switch of this context (None removes it)."""
        L = self._L
        rc = L.dsm_set_debug_option(self._handle, key.encode(), None if value is None else str(value).encode())
        if rc != 0:
            raise DsmError("dsm_set_debug_option(%s) failed (%d)" % (key, rc))
        if value is None:
            self._debug.pop(key, None)
        else:
            self._debug[key] = str(value)

    def close(self):
        if self._handle:
            self._L.dsm_ctx_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise DsmError("dsm error %d: %s" % (rc, self._L.dsm_last_error(self._h).decode()))

    def sync(self):
        self._chk(self._L.dsm_sync(self._h))

    def append_images(self, descriptors, keypoints=None, cameras=None):
        """dsm_append_images: adds images behind the resident ones (indices continue), uploading only the new rows."""
        self.set_images(descriptors, keypoints, cameras, _append=True)

    def set_images(self, descriptors, keypoints=None, cameras=None, _append=False):
        """descriptors: list of (n_i,128) uint8; keypoints: list of (n_i,>=2) float32; cameras: list of Camera."""
        n = len(descriptors)
        descs = [np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, 128) for d in descriptors]
        nf = np.array([d.shape[0] for d in descs], dtype=np.uint32)
        vp = ctypes.c_void_p
        dptr = (vp * max(n, 1))(*[d.ctypes.data for d in descs])
        kptr = None
        stride = 0
        kps = None
        if keypoints is not None:
            kps = [np.ascontiguousarray(k, dtype=np.float32) for k in keypoints]
            kps = [k.reshape(-1, k.shape[-1] if k.ndim == 2 else 2) for k in kps]
            stride = kps[0].shape[1] if n else 2
            for k, d in zip(kps, descs):
                assert k.shape[0] == d.shape[0] and k.shape[1] == stride
            kptr = (vp * max(n, 1))(*[k.ctypes.data for k in kps])
        cptr = None
        if cameras is not None:
            cptr = (Camera * max(n, 1))(*cameras)
        fn = self._L.dsm_append_images if _append else self._L.dsm_set_images
        fn.argtypes = self._L.dsm_set_images.argtypes
        self._chk(fn(self._h, n, nf.ctypes.data_as(u32p), dptr, kptr, stride, cptr))
        self._keep = (descs, kps)

    def match_pairs(self, pairs, options=None):
        options = options or default_match_options()
        p = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        self.n_pairs = p.shape[0]
        self._chk(self._L.dsm_match_pairs(self._h, self.n_pairs, p.ctypes.data_as(u32p), ctypes.byref(options)))

    def set_matches(self, pairs, matches_per_pair):
        """dsm_set_matches: installs given FeatureMatches for the pair list (the verify-only / resume path of
        SiftFeatureMatcher::Match, matching.cc:806-812) instead of running the matcher."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        off = np.zeros(len(pairs) + 1, dtype=np.uint64)
        for k, m in enumerate(matches_per_pair):
            off[k + 1] = off[k] + len(m)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(m, dtype=np.uint32).reshape(-1, 2) for m in matches_per_pair] +
                                                   [np.zeros((1, 2), np.uint32)]), dtype=np.uint32)
        self._L.dsm_set_matches.argtypes = [ctypes.c_void_p, ctypes.c_uint32, u32p, ctypes.POINTER(ctypes.c_uint64), u32p]
        self._chk(self._L.dsm_set_matches(self._h, len(pairs), pairs.ctypes.data_as(u32p),
                                        off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), flat.ctypes.data_as(u32p)))
        self.n_pairs = len(pairs)

    def match_counts(self):
        c = np.zeros(max(self.n_pairs, 1), dtype=np.uint32)
        self._chk(self._L.dsm_get_match_counts(self._h, c.ctypes.data))
        return c[:self.n_pairs]

    def matches(self):
        """Returns (offsets[n_pairs+1], matches[total,2])."""
        offs = np.zeros(self.n_pairs + 1, dtype=np.uint64)
        self._chk(self._L.dsm_get_matches(self._h, offs.ctypes.data, None, 0))
        total = int(offs[-1])
        m = np.zeros((max(total, 1), 2), dtype=np.uint32)
        self._chk(self._L.dsm_get_matches(self._h, None, m.ctypes.data, total))
        return offs, m[:total]

    def match_sift_features(self, desc1, desc2, options=None):
        """MatchSiftFeaturesCPU-shaped leaf, /root/reference/src/feature/sift.h:214-217."""
        options = options or default_match_options()
        d1 = np.ascontiguousarray(desc1, dtype=np.uint8).reshape(-1, 128)
        d2 = np.ascontiguousarray(desc2, dtype=np.uint8).reshape(-1, 128)
        out = np.zeros((max(d1.shape[0], 1), 2), dtype=np.uint32)
        n = ctypes.c_uint32(0)
        self._chk(self._L.dsm_match_sift_features(self._h, ctypes.byref(options), d1.ctypes.data_as(u8p), d1.shape[0],
                                                d2.ctypes.data_as(u8p), d2.shape[0], out.ctypes.data_as(u32p),
                                                ctypes.byref(n)))
        return out[:n.value].copy()

    def verify_pairs(self, options=None, seeds=None, user_seed=0, stage_filter=True):
        options = options or default_two_view_options()
        sp = None
        if seeds is not None:
            self._seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
            assert len(self._seeds) == self.n_pairs
            sp = self._seeds.ctypes.data_as(u32p)
        self._chk(self._L.dsm_verify_pairs(self._h, ctypes.byref(options), sp, user_seed, int(bool(stage_filter))))

    def two_view_geometries(self):
        arr = (TwoViewGeometry * max(self.n_pairs, 1))()
        self._chk(self._L.dsm_get_two_view_geometries(self._h, ctypes.addressof(arr)))
        return list(arr)[:self.n_pairs]

    def inlier_matches(self):
        offs = np.zeros(self.n_pairs + 1, dtype=np.uint64)
        self._chk(self._L.dsm_get_inlier_matches(self._h, offs.ctypes.data, None, 0))
        total = int(offs[-1])
        m = np.zeros((max(total, 1), 2), dtype=np.uint32)
        self._chk(self._L.dsm_get_inlier_matches(self._h, None, m.ctypes.data, total))
        return offs, m[:total]

    def verify_kernel_time(self):
        ms = ctypes.c_double(0)
        self._chk(self._L.dsm_get_verify_kernel_time(self._h, ctypes.byref(ms)))
        return ms.value

    def estimate_two_view_geometry(self, cam1, pts1, cam2, pts2, matches, options=None, seed=0):
        """TwoViewGeometry::Estimate-shaped leaf, /root/reference/src/estimators/two_view_geometry.h:180-184."""
        options = options or default_two_view_options()
        p1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
        p2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
        m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 2)
        out = TwoViewGeometry()
        inl = np.zeros((max(len(m), 1), 2), dtype=np.uint32)
        dp = ctypes.POINTER(ctypes.c_double)
        self._chk(self._L.dsm_estimate_two_view_geometry(self._h, ctypes.byref(cam1), p1.ctypes.data_as(dp), len(p1),
                                                       ctypes.byref(cam2), p2.ctypes.data_as(dp), len(p2),
                                                       m.ctypes.data_as(u32p), len(m), ctypes.byref(options), seed,
                                                       ctypes.byref(out), inl.ctypes.data_as(u32p)))
        return out, inl[:out.num_inliers].copy()

    def debug_sample_sequence(self, seed, k, total, n_draws):
        out = np.zeros((n_draws, k), dtype=np.uint32)
        self._chk(self._L.dsm_debug_sample_sequence(self._h, seed, k, total, n_draws, out.ctypes.data_as(u32p)))
        return out

    def debug_image_to_world(self, cam, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        out = np.zeros_like(xy)
        dp = ctypes.POINTER(ctypes.c_double)
        self._chk(self._L.dsm_debug_image_to_world(self._h, ctypes.byref(cam), len(xy), xy.ctypes.data_as(dp), out.ctypes.data_as(dp)))
        return out

    def match_kernel_time(self):
        ms = ctypes.c_double(0)
        n = ctypes.c_uint32(0)
        self._chk(self._L.dsm_get_match_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def guided_match_pairs(self, match_options=None, options=None, stage_filter=False):
        """dsm_guided_match_pairs: replaces the inlier matches of the verified pairs by guided matches."""
        mo = match_options if match_options is not None else default_match_options()
        to = options if options is not None else default_two_view_options()
        self._chk(self._L.dsm_guided_match_pairs(self._h, ctypes.byref(mo), ctypes.byref(to), 1 if stage_filter else 0))

    # ---- vocabulary-tree retrieval (candidate pairs)
    def retrieval_set_vocabulary(self, words, projection, thresholds):
        self._voc = (np.ascontiguousarray(words, np.uint8).reshape(-1, 128), np.ascontiguousarray(projection, np.float32).reshape(64, 128),
                     np.ascontiguousarray(thresholds, np.float32).reshape(-1, 64))
        assert self._voc[2].shape[0] == self._voc[0].shape[0]
        v = Vocabulary(num_words=self._voc[0].shape[0], reserved=0, words=self._voc[0].ctypes.data, projection=self._voc[1].ctypes.data,
                       thresholds=self._voc[2].ctypes.data)
        self._chk(self._L.dsm_retrieval_set_vocabulary(self._h, ctypes.byref(v)))

    def debug_verify_counters(self):
        """dsm_debug_verify_counters: 16 statistics counters of the last verify call (DSM_VERIFY_DEBUG / DSM_SCORE_PREFILTER=check)."""
        out = np.zeros(16, np.uint32)
        L = self._L
        L.dsm_debug_verify_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(L.dsm_debug_verify_counters(self._h, out.ctypes.data))
        return out

    def retrieval_set_word_ids(self, index_ids, query_ids):
        """dsm_retrieval_set_word_ids: the caller's word ids (the reference's FLANN answer) instead of the device's exact
        search; index_ids [features], query_ids [features, k].  (None, None): exact search again."""
        L = self._L
        L.dsm_retrieval_set_word_ids.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        if index_ids is None:
            self._chk(L.dsm_retrieval_set_word_ids(self._h, None, 0, None))
            return
        a = np.ascontiguousarray(index_ids, np.int32).reshape(-1)
        b = np.ascontiguousarray(query_ids, np.int32).reshape(len(a), -1)
        self._chk(L.dsm_retrieval_set_word_ids(self._h, a.ctypes.data, b.shape[1], b.ctypes.data))

    def retrieval_index(self):
        self._chk(self._L.dsm_retrieval_index(self._h))

    def retrieval_query(self, n_images, num_neighbors=5, max_num_images=100):
        """Returns a list (per query image, in dsm_set_images order) of (image_idx [c], scores [c]) in retrieval order."""
        cnt = np.zeros(n_images, np.uint32)
        idx = np.zeros((n_images, max_num_images), np.uint32)
        sc = np.zeros((n_images, max_num_images), np.float32)
        self._chk(self._L.dsm_retrieval_query(self._h, num_neighbors, max_num_images, cnt.ctypes.data, idx.ctypes.data, sc.ctypes.data))
        return [(idx[q, :cnt[q]].copy(), sc[q, :cnt[q]].copy()) for q in range(n_images)]

    def retrieval_matches(self, query_result, num_neighbors=5, max_num_images=100):
        """dsm_retrieval_matches + dsm_get_retrieval_matches for the retrieved lists of retrieval_query: (offsets [n+1] uint64,
        tuples [total, 5] uint32 = query feature, image, database feature, word << 8 | Hamming distance, entry position)."""
        n = len(query_result)
        cnt = np.array([len(r[0]) for r in query_result], np.uint32)
        idx = np.zeros((n, max_num_images), np.uint32)
        for q, r in enumerate(query_result):
            idx[q, :len(r[0])] = r[0]
        offs = np.zeros(n + 1, np.uint64)
        L = self._L
        L.dsm_retrieval_matches.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.dsm_get_retrieval_matches.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        self._chk(L.dsm_retrieval_matches(self._h, num_neighbors, max_num_images, cnt.ctypes.data, idx.ctypes.data, offs.ctypes.data))
        total = int(offs[-1])
        tup = np.zeros((max(total, 1), 5), np.uint32)
        self._chk(L.dsm_get_retrieval_matches(self._h, tup.ctypes.data, total))
        return offs, tup[:total]

    def retrieval_idf(self, num_words):
        out = np.zeros(num_words, np.float32)
        L = self._L
        L.dsm_get_retrieval_idf.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        self._chk(L.dsm_get_retrieval_idf(self._h, out.ctypes.data, num_words))
        return out

    def retrieval_debug_word_ids(self, image, n_feats, k):
        out = np.zeros((max(n_feats, 1), k), np.int32)
        self._chk(self._L.dsm_retrieval_debug_word_ids(self._h, image, k, out.ctypes.data))
        return out[:n_feats]

    def retrieval_time(self):
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        self._chk(self._L.dsm_get_retrieval_time(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def view_graph_filter_cycles(self, pairs, qvecs, max_loop_error_degrees=5.0):
        """ViewGraph::FilterViewGraphCyclesByRotation over (pairs, qvecs): returns (keep [n] bool, number of triplets)."""
        p = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        q = np.ascontiguousarray(qvecs, np.float64).reshape(-1, 4)
        assert len(p) == len(q)
        keep = np.zeros(max(len(p), 1), np.uint8)
        nt = ctypes.c_uint64(0)
        self._chk(self._L.dsm_view_graph_filter_cycles(self._h, len(p), p.ctypes.data, q.ctypes.data, max_loop_error_degrees,
                                                     keep.ctypes.data, ctypes.addressof(nt)))
        return keep[:len(p)].astype(bool), nt.value

    def device_info(self):
        d = DeviceInfo()
        self._chk(self._L.dsm_get_device_info(self._h, ctypes.byref(d)))
        return d

    def match_gather_time(self):
        ms = ctypes.c_double(0)
        self._chk(self._L.dsm_get_match_gather_time(self._h, ctypes.byref(ms)))
        return ms.value

    def match_tail_time(self):
        ms = ctypes.c_double()
        self._chk(self._L.dsm_get_match_tail_time(self._h, ctypes.byref(ms)))
        return ms.value

    def match_resolve_time(self):
        ms = ctypes.c_double(0)
        self._chk(self._L.dsm_get_match_resolve_time(self._h, ctypes.byref(ms)))
        return ms.value


_gather_lib = None


def gather_lib():
    """libdagsfm_gather.so (include/dagsfm_gather.h); maps librccl, so it is loaded on first use only."""
    global _gather_lib
    if _gather_lib is None:
        if not os.path.exists(GATHER_LIB_PATH):
            raise DsmError("%s is missing: make -C dagsfm_amd/csrc" % GATHER_LIB_PATH)
        lib()  # the companion links the product library: resolve it to the in-tree one first
        G = ctypes.CDLL(GATHER_LIB_PATH)
        vp = ctypes.c_void_p
        G.dsm_gather_create.argtypes = [ctypes.POINTER(vp), ctypes.c_uint32, ctypes.POINTER(vp)]
        G.dsm_gather_destroy.argtypes = [vp]
        G.dsm_gather_destroy.restype = None
        G.dsm_gather_last_error.argtypes = [vp]
        G.dsm_gather_last_error.restype = ctypes.c_char_p
        G.dsm_gather_match_graph.argtypes = [vp, u32p, ctypes.c_int32]
        G.dsm_gather_sizes.argtypes = [vp, u64p, u64p, u64p]
        G.dsm_gather_fetch.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, vp, vp]
        G.dsm_gather_device_arrays.argtypes = [vp, ctypes.c_uint32] + [ctypes.POINTER(vp)] * 5
        G.dsm_gather_time.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        _gather_lib = G
    return _gather_lib


class Gather:
    """dsm_gather over product contexts on distinct devices of this process (one context: a one-rank communicator)."""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)
        assert all(not c.check for c in self.ctxs), "the companion library links the product build"
        self._G = gather_lib()
        arr = (ctypes.c_void_p * len(self.ctxs))(*[c._h for c in self.ctxs])
        self._g = ctypes.c_void_p()
        rc = self._G.dsm_gather_create(arr, len(self.ctxs), ctypes.byref(self._g))
        if rc != 0:
            raise DsmError("dsm_gather_create failed (%d)" % rc)

    def _chk(self, rc):
        if rc != 0:
            raise DsmError("dsm_gather error %d: %s" % (rc, self._G.dsm_gather_last_error(self._g).decode()))

    def close(self):
        if self._g:
            self._G.dsm_gather_destroy(self._g)
            self._g = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match_graph(self, n_pairs, with_geometry=True, rank=0):
        """Assembles the shares and fetches the graph from `rank`'s device: (match offsets, matches, records, inlier offsets, inlier matches)."""
        npairs = np.ascontiguousarray(n_pairs, np.uint32)
        assert len(npairs) == len(self.ctxs)
        self._chk(self._G.dsm_gather_match_graph(self._g, npairs.ctypes.data_as(u32p), 1 if with_geometry else 0))
        N, M, I = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(self._G.dsm_gather_sizes(self._g, ctypes.byref(N), ctypes.byref(M), ctypes.byref(I)))
        moff = np.zeros(N.value + 1, np.uint64)
        m = np.zeros((max(M.value, 1), 2), np.uint32)
        if not with_geometry:
            self._chk(self._G.dsm_gather_fetch(self._g, rank, moff.ctypes.data, m.ctypes.data, None, None, None))
            return moff, m[:M.value]
        tv = (TwoViewGeometry * max(N.value, 1))()
        ioff = np.zeros(N.value + 1, np.uint64)
        im = np.zeros((max(I.value, 1), 2), np.uint32)
        self._chk(self._G.dsm_gather_fetch(self._g, rank, moff.ctypes.data, m.ctypes.data, ctypes.addressof(tv), ioff.ctypes.data, im.ctypes.data))
        return moff, m[:M.value], list(tv)[:N.value], ioff, im[:I.value]

    def time_ms(self):
        ms = ctypes.c_double()
        self._chk(self._G.dsm_gather_time(self._g, ctypes.byref(ms)))
        return ms.value
