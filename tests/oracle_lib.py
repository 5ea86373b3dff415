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
