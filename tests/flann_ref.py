"""ctypes binding of oracle/_ref/libflann_ref.so -- the reference's OWN visual-word search (FLANN, vendored under
/root/reference/lib/FLANN) compiled from where it lies by `make -C oracle ref` (oracle/ref_flann_shim.cpp).
Test infrastructure only; None where the library was never built (no /root/reference on that machine)."""
import ctypes
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libflann_ref.so")
LINEAR, KDTREE, KMEANS = 0, 1, 2
_lib = None


def load():
    global _lib
    if _lib is None and os.path.exists(PATH):
        L = ctypes.CDLL(PATH)
        vp = ctypes.c_void_p
        L.flann_ref_seed.argtypes = [ctypes.c_uint]
        L.flann_ref_quantize.argtypes = [vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        L.flann_ref_build.restype = vp
        L.flann_ref_build.argtypes = [vp, ctypes.c_uint32, ctypes.c_float]
        L.flann_ref_build_forced.restype = vp
        L.flann_ref_build_forced.argtypes = [vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.flann_ref_save_append.restype = ctypes.c_long
        L.flann_ref_save_append.argtypes = [vp, ctypes.c_char_p]
        L.flann_ref_load.restype = vp
        L.flann_ref_load.argtypes = [vp, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_long, ctypes.POINTER(ctypes.c_long)]
        L.flann_ref_destroy.argtypes = [vp]
        L.flann_ref_set_search.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.flann_ref_knn.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
        L.flann_ref_algorithm.argtypes = [vp]
        _lib = L
    return _lib


class Index:
    def __init__(self, handle, words):
        self.h, self.words = handle, words

    @staticmethod
    def build(words, target_precision=0.9):
        w = np.ascontiguousarray(words, np.uint8)
        return Index(load().flann_ref_build(w.ctypes.data, len(w), target_precision), w)

    @staticmethod
    def build_forced(words, algorithm, p1=4, p2=5, autotuned_checks=32, seed=1):
        w = np.ascontiguousarray(words, np.uint8)
        load().flann_ref_seed(seed)
        return Index(load().flann_ref_build_forced(w.ctypes.data, len(w), algorithm, p1, p2, autotuned_checks), w)

    @staticmethod
    def load(words, path, offset):
        w = np.ascontiguousarray(words, np.uint8)
        end = ctypes.c_long(-1)
        h = load().flann_ref_load(w.ctypes.data, len(w), path.encode(), offset, ctypes.byref(end))
        if not h:
            raise RuntimeError("flann loadIndex failed")
        ix = Index(h, w)
        ix.end_offset = end.value
        return ix

    def save_append(self, path):
        return load().flann_ref_save_append(self.h, path.encode())

    def algorithm(self):
        return load().flann_ref_algorithm(self.h)

    def knn(self, descriptors, k, num_checks=256, cores=1, with_dists=False):
        d = np.ascontiguousarray(descriptors, np.uint8)
        load().flann_ref_set_search(self.h, num_checks, cores)
        ids = np.zeros((len(d), k), np.int32)
        dists = np.zeros((len(d), k), np.float32)
        load().flann_ref_knn(self.h, d.ctypes.data, len(d), k, ids.ctypes.data, dists.ctypes.data)
        return (ids, dists) if with_dists else ids

    def close(self):
        if self.h:
            load().flann_ref_destroy(self.h)
            self.h = None


def write_reference_vocabulary(path, words, projection, thresholds, index, rng=None, with_entries=False):
    """A vocabulary-tree file exactly as VisualIndex<>::Write leaves it (retrieval/visual_index.h:586-614): the words, the
    REAL flann saveIndex output (appended to the file by the reference's own code), the inverted index
    (inverted_index.h:383-420 / inverted_file.h:394-411).  Returns (offset of the FLANN index, offset of the inverted index)."""
    w = np.ascontiguousarray(words, np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", w.shape[0], 128))
        f.write(w.tobytes())
    begin = os.path.getsize(path)
    end = index.save_append(path)
    with open(path, "ab") as f:
        f.write(struct.pack("<ii", w.shape[0], 64))
        f.write(np.ascontiguousarray(projection, np.float32).tobytes())
        n_img = 0
        for k in range(w.shape[0]):
            f.write(struct.pack("<Bf", 3, 0.25 * k))
            f.write(np.ascontiguousarray(thresholds[k], np.float32).tobytes())
            ne = int(rng.integers(0, 3)) if with_entries else 0
            f.write(struct.pack("<I", ne))
            for _ in range(ne):
                f.write(struct.pack("<iiffffQ", 7, 3, 1.0, 2.0, 3.0, 0.5, 0xDEADBEEF))
                n_img = 1
        f.write(struct.pack("<i", n_img))
        for _ in range(n_img):
            f.write(struct.pack("<if", 7, 1.5))
    return begin, end
