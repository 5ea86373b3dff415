"""SURVEY 8f-2 / VERDICT r03 next 1: the retrieval row pinned to the reference's OWN visual-word search.

VisualIndex::FindWordIds (/root/reference/src/retrieval/visual_index.h:695-738) searches a flann::AutotunedIndex LOADED from
the vocabulary file (:564-574); FLANN is vendored under /root/reference/lib/FLANN and compiles here
(oracle/ref_flann_shim.cpp -> oracle/_ref/libflann_ref.so, `make -C oracle ref`).  Checked, CPU only:

  * a vocabulary file whose middle section is a REAL saveIndex output (linear / kd-trees / k-means; FLANN's header, LZ4
    blocks): the host shim's reader walks FLANN's archive framing and lands on the byte loadIndex stops at;
  * the host shim's FLANN-compatible search (dagsfm_amd/host/flann_index.cc: LZ4 decoding, archive layout, the kd-tree /
    k-means / linear searches, KNNSimpleResultSet, L2<uint8_t> in float) returns the reference's word ids AND float
    distances bit for bit -- against the compiled reference where it is available, against the committed golden files
    (tools/make_flann_golden.py) everywhere;
  * how far the product's default (exact nearest words on the device) is from the reference's approximate answer is a
    measurement, kept in profiles/r04_flann_agreement.json (tools/flann_agreement.py)."""
import ctypes
import os
import struct

import numpy as np
import pytest

from tests import flann_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
u64p = ctypes.POINTER(ctypes.c_uint64)


def _host():
    L = ctypes.CDLL(os.path.join(ROOT, "dagsfm_amd", "libdagsfm_host.so"))
    L.dsm_host_read_vocabulary.restype = ctypes.c_uint32
    L.dsm_host_read_vocabulary.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    L.dsm_host_vocabulary_index_range.restype = ctypes.c_uint32
    L.dsm_host_vocabulary_index_range.argtypes = [ctypes.c_char_p, u64p, u64p, ctypes.POINTER(ctypes.c_int)]
    L.dsm_host_flann_find_word_ids.restype = ctypes.c_int
    L.dsm_host_flann_find_word_ids.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, u64p]
    return L


def _product_search(L, path, queries, k, checks, threads=1):
    q = np.ascontiguousarray(queries, np.uint8)
    ids = np.zeros((len(q), k), np.int32)
    dists = np.zeros((len(q), k), np.float32)
    end = ctypes.c_uint64(0)
    algo = L.dsm_host_flann_find_word_ids(path.encode(), q.ctypes.data, len(q), k, checks, threads, ids.ctypes.data, dists.ctypes.data,
                                          ctypes.byref(end))
    return algo, ids, dists, end.value


def _range(L, path):
    b, e, fr = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int(-1)
    n = L.dsm_host_vocabulary_index_range(path.encode(), ctypes.byref(b), ctypes.byref(e), ctypes.byref(fr))
    return n, b.value, e.value, fr.value


@pytest.mark.parametrize("name,algo", [("linear", 0), ("kdtree", 1), ("kmeans", 2)])
def test_golden_vocabulary_files_with_real_flann_indices(name, algo):
    """Committed files written by the reference's own saveIndex (tools/make_flann_golden.py): the reader finds the inverted
    index at the offset the reference's loadIndex reported, and the FLANN-compatible search returns what the reference's
    knnSearch over the loaded index returned -- ids and float distances, num_checks 32 and 256."""
    L = _host()
    exp = np.load(os.path.join(GOLDEN, "vocab_flann_expected.npz"))
    path = os.path.join(GOLDEN, "vocab_flann_%s.bin" % name)
    n, begin, end, framed = _range(L, path)
    assert n == len(exp["words"]) and framed == 1 and [begin, end] == list(exp[name + "_range"])
    w = np.zeros_like(exp["words"])
    p = np.zeros((64, 128), np.float32)
    t = np.zeros((n, 64), np.float32)
    assert L.dsm_host_read_vocabulary(path.encode(), w.ctypes.data, p.ctypes.data, t.ctypes.data, n) == n
    assert (w == exp["words"]).all() and (p == exp["projection"]).all() and (t == exp["thresholds"]).all()
    for checks in (32, 256):
        for threads in (1, 3):
            a, ids, dists, stop = _product_search(L, path, exp["queries"], 5, checks, threads)
            assert a == algo and stop == end
            assert (ids == exp["%s_ids_%d" % (name, checks)]).all()
            assert (dists == exp["%s_dists_%d" % (name, checks)]).all()


def _need_ref():
    if flann_ref.load() is None:
        pytest.skip("oracle/_ref/libflann_ref.so was not built (no /root/reference on this machine)")


def _sift_like(rng, n):
    d = rng.random((n, 128)) ** 2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.minimum(np.round(d * 512), 255).astype(np.uint8)


@pytest.mark.parametrize("algo,p1,p2,n_words", [(flann_ref.KDTREE, 1, 0, 700), (flann_ref.KDTREE, 8, 0, 3000), (flann_ref.KMEANS, 16, 5, 3000),
                                                (flann_ref.KMEANS, 3, 1, 500), (flann_ref.KMEANS, 32, 15, 9000), (flann_ref.LINEAR, 0, 0, 300),
                                                (flann_ref.KDTREE, 4, 0, 20000)])
def test_flann_compatible_search_equals_the_references_flann(tmp_path, algo, p1, p2, n_words):
    """Fresh indices built, written and re-loaded by the reference's FLANN; clustered words (duplicates included, so that
    equal distances occur), queries near and far; k = 1 (VisualIndex::Add) and 5 (Query); several num_checks.  A 20 000-word
    kd-tree archive spans several 64 KiB LZ4 blocks with back-references across them."""
    _need_ref()
    L = _host()
    rng = np.random.default_rng(n_words + 31 * algo + p1)
    centers = _sift_like(rng, 40)
    words = np.clip(centers[rng.integers(0, 40, n_words)].astype(np.int32) + rng.integers(-12, 13, (n_words, 128)), 0, 255).astype(np.uint8)
    words[5] = words[6]  # exact duplicates: equal distances
    proj = rng.standard_normal((64, 128)).astype(np.float32)
    thr = rng.standard_normal((n_words, 64)).astype(np.float32)
    ix = flann_ref.Index.build_forced(words, algo, p1, p2, autotuned_checks=17, seed=n_words)
    path = str(tmp_path / "vocab.bin")
    begin, end = flann_ref.write_reference_vocabulary(path, words, proj, thr, ix, rng, with_entries=True)
    ix.close()
    ref = flann_ref.Index.load(words, path, begin)
    assert ref.end_offset == end and ref.algorithm() == algo
    n, b, e, framed = _range(L, path)
    assert (n, b, e, framed) == (n_words, begin, end, 1)
    queries = np.concatenate([np.clip(centers[rng.integers(0, 40, 300)].astype(np.int32) + rng.integers(-20, 21, (300, 128)), 0, 255),
                              words[:40], rng.integers(0, 256, (60, 128))]).astype(np.uint8)
    for k in (1, 5):
        for checks in (1, 32, 256, -2):
            rids, rd = ref.knn(queries, k, num_checks=checks, with_dists=True)
            a, ids, dists, stop = _product_search(L, path, queries, k, checks, threads=2)
            assert a == algo and stop == end
            assert (ids == rids).all(), (k, checks, int((ids != rids).sum()))
            assert (dists == rd).all()
    ref.close()


def test_autotuned_build_round_trip(tmp_path):
    """AutotunedIndex::buildIndex as VisualIndex::Build calls it (target_precision only): whatever the tuner picks on this
    machine (its decision rests on wall-clock timings), the file it writes is read back to the same answers."""
    _need_ref()
    L = _host()
    rng = np.random.default_rng(5)
    words = _sift_like(rng, 1200)
    flann_ref.load().flann_ref_seed(3)
    ix = flann_ref.Index.build(words, 0.9)
    path = str(tmp_path / "vocab.bin")
    begin, end = flann_ref.write_reference_vocabulary(path, words, rng.standard_normal((64, 128)).astype(np.float32),
                                                      rng.standard_normal((len(words), 64)).astype(np.float32), ix)
    algo = ix.algorithm()
    ix.close()
    ref = flann_ref.Index.load(words, path, begin)
    queries = _sift_like(rng, 200)
    a, ids, dists, stop = _product_search(L, path, queries, 5, 256)
    rids, rd = ref.knn(queries, 5, num_checks=256, with_dists=True)
    assert a == algo == ref.algorithm() and stop == end == ref.end_offset
    assert (ids == rids).all() and (dists == rd).all()
    ref.close()


def test_damaged_flann_sections_are_refused(tmp_path):
    """A vocabulary file whose FLANN section is cut, or whose block sizes lie, is an error -- never an over-read."""
    L = _host()
    src = open(os.path.join(GOLDEN, "vocab_flann_kmeans.bin"), "rb").read()
    exp = np.load(os.path.join(GOLDEN, "vocab_flann_expected.npz"))
    begin, end = [int(v) for v in exp["kmeans_range"]]
    q = exp["queries"][:4]
    rng = np.random.default_rng(0)
    n_bad = 0
    for trial in range(60):
        b = bytearray(src)
        pos = begin + int(rng.integers(0, end - begin))
        if trial % 3 == 0:
            b[pos] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 1:
            b[pos:pos + 8] = struct.pack("<Q", int(rng.integers(0, 1 << 40)))
        else:
            del b[pos:pos + int(rng.integers(1, 64))]
        path = str(tmp_path / ("bad%d.bin" % trial))
        open(path, "wb").write(bytes(b))
        algo, ids, dists, stop = _product_search(L, path, q, 5, 32)  # must return, with an error or with in-range ids
        if algo < 0:
            n_bad += 1
        else:
            assert ((ids >= 0) & ((ids < len(exp["words"])) | (ids == 2147483647))).all()
    assert n_bad > 20
