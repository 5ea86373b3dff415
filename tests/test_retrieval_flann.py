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


# ------------------------------------------------------------------------------------------- the product's word_search = flann
def _flann_scene(tmp_path, rng, n_img, feats, n_words, algo, p1, p2):
    from dagsfm_amd import synthetic
    scene = synthetic.Scene(n_img, feats, seed=17)
    ims = [scene.image(i) for i in range(n_img)]
    desc = np.concatenate([im[0] for im in ims])
    words = desc[rng.choice(len(desc), n_words, replace=False)].copy()
    proj = rng.standard_normal((64, 128)).astype(np.float32)
    # per-word thresholds around the projected words themselves, so that signatures are not all alike
    thr = (proj @ words.astype(np.float32).T).T.astype(np.float32)
    ix = flann_ref.Index.build_forced(words, algo, p1, p2, autotuned_checks=32, seed=9)
    path = str(tmp_path / "vocab_tree.bin")
    begin, end = flann_ref.write_reference_vocabulary(path, words, proj, thr, ix)
    ix.close()
    return ims, words, proj, thr, path, begin


@pytest.mark.gpu
@pytest.mark.parametrize("algo,p1,p2", [(flann_ref.KDTREE, 4, 0), (flann_ref.KMEANS, 8, 3)])
def test_word_search_flann_equals_the_oracle_over_the_references_flann(tmp_path, dsm, algo, p1, p2):
    """VocabSimilarityGraph::Run with word_search = flann over a database.db and a vocabulary file carrying a REAL FLANN
    tree: the candidate pairs and scores equal the oracle's whose every word search is the reference's own
    flann::AutotunedIndex::knnSearch over the loaded index (oracle_retrieval_set_word_search <- libflann_ref.so) -- and
    differ from the exact-search run, which is the point of the mode."""
    _need_ref()
    from tests import dbutil, oracle_lib
    rng = np.random.default_rng(77 + algo)
    n_img, feats, n_words, k, max_images, checks = 10, 400, 1500, 5, 6, 24
    ims, words, proj, thr, vpath, begin = _flann_scene(tmp_path, rng, n_img, feats, n_words, algo, p1, p2)
    dpath = str(tmp_path / "database.db")
    dbutil.create(dpath, [(im[0], im[1]) for im in ims])
    L = _host()
    L.dsm_host_vocab_candidate_pairs4.restype = ctypes.c_int64
    L.dsm_host_vocab_candidate_pairs4.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    res = {}
    for mode in (1, 2, 3, 0):  # FLANN on the device, FLANN on host threads, auto (= FLANN: the file carries an index), exact
        pairs = np.zeros((1000, 2), np.uint32)
        scores = np.zeros(1000, np.float32)
        n = L.dsm_host_vocab_candidate_pairs4(dpath.encode(), vpath.encode(), max_images, k, -1, 0, mode, checks, pairs.ctypes.data,
                                              scores.ctypes.data, 1000)
        assert n > 0
        res[mode] = ([tuple(x) for x in pairs[:n]], scores[:n].copy())
    ref = flann_ref.Index.load(words, vpath, begin)
    orc = oracle_lib.RetrievalOracle(words, proj, thr)
    orc.use_flann(ref, checks)
    for i, im in enumerate(ims):
        orc.add(i, im[0])
    orc.prepare()
    exp_pairs, exp_scores = [], []
    for q, im in enumerate(ims):
        ids, sc = orc.query(im[0], k, max_images)
        for d, s in zip(ids, sc):
            if q < int(d):
                exp_pairs.append((q + 1, int(d) + 1))
                exp_scores.append(np.float32(s) * np.float32(1e3))
    for mode in (1, 2, 3):
        assert res[mode][0] == exp_pairs and (res[mode][1] == np.array(exp_scores, np.float32)).all(), mode
    # the approximate search at 24 checks is not the exact one: the two modes score differently
    assert res[0][0] != res[1][0] or not np.array_equal(res[0][1], res[1][1])
    # and the device entry point itself, fed the reference's ids directly
    dsm.set_images([im[0] for im in ims])
    dsm.retrieval_set_vocabulary(words, proj, thr)
    alld = np.concatenate([im[0] for im in ims])
    dsm.retrieval_set_word_ids(ref.knn(alld, 1, num_checks=checks), ref.knn(alld, k, num_checks=checks))
    dsm.retrieval_index()
    got = dsm.retrieval_query(n_img, k, max_images)
    dsm.retrieval_set_word_ids(None, None)
    for q, im in enumerate(ims):
        ids, sc = orc.query(im[0], k, max_images)
        assert list(got[q][0]) == list(ids) and (got[q][1] == sc).all()
    ref.close()


# ------------------------------------------------------------------------------------------- the FLANN search ON THE DEVICE (round 5)
def _device_search(L, path, queries, k, checks):
    L.dsm_host_flann_device_search.restype = ctypes.c_int
    L.dsm_host_flann_device_search.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    q = np.ascontiguousarray(queries, np.uint8)
    ids = np.zeros((len(q), k), np.int32)
    dists = np.zeros((len(q), k), np.float32)
    ms = ctypes.c_double(0)
    algo = L.dsm_host_flann_device_search(path.encode(), 0, q.ctypes.data, len(q), k, checks, ids.ctypes.data, dists.ctypes.data, ctypes.byref(ms))
    return algo, ids, dists, ms.value


@pytest.mark.gpu
@pytest.mark.parametrize("name,algo", [("linear", 0), ("kdtree", 1), ("kmeans", 2)])
def test_device_flann_search_equals_the_references_answers_on_the_goldens(name, algo):
    """csrc/flann_search.hip against the committed answers of the reference's own knnSearch over the loaded index
    (tools/make_flann_golden.py): ids AND float distances, k = 5, num_checks 32 and 256; and against the host restatement for
    k = 1 (VisualIndex::Add's search) and k = 8, where the goldens hold no answer."""
    L = _host()
    exp = np.load(os.path.join(GOLDEN, "vocab_flann_expected.npz"))
    path = os.path.join(GOLDEN, "vocab_flann_%s.bin" % name)
    for checks in (32, 256):
        a, ids, dists, ms = _device_search(L, path, exp["queries"], 5, checks)
        assert a == algo
        assert (ids == exp["%s_ids_%d" % (name, checks)]).all(), int((ids != exp["%s_ids_%d" % (name, checks)]).sum())
        assert (dists == exp["%s_dists_%d" % (name, checks)]).all()
        for k in (1, 8):
            a, ids, dists, ms = _device_search(L, path, exp["queries"], k, checks)
            ha, hids, hdists, _ = _product_search(L, path, exp["queries"], k, checks, 2)
            assert a == ha == algo and (ids == hids).all() and (dists == hdists).all()


@pytest.mark.gpu
@pytest.mark.parametrize("algo,p1,p2,n_words", [(flann_ref.KDTREE, 1, 0, 700), (flann_ref.KDTREE, 8, 0, 3000), (flann_ref.KMEANS, 16, 5, 3000),
                                                (flann_ref.KMEANS, 3, 1, 500), (flann_ref.KMEANS, 32, 15, 9000), (flann_ref.LINEAR, 0, 0, 300),
                                                (flann_ref.KDTREE, 4, 0, 20000)])
def test_device_flann_search_equals_the_references_flann_on_fresh_indices(tmp_path, algo, p1, p2, n_words):
    """Fresh indices built, written and re-loaded by the reference's FLANN (oracle/_ref/libflann_ref.so): clustered words with exact
    duplicates (equal distances: the heap's and the result set's tie order matters), queries near and far, more queries than one
    wave holds; k = 1 and 5; num_checks 1, 32, 256 and the index's own autotuned estimate."""
    _need_ref()
    L = _host()
    rng = np.random.default_rng(n_words + 31 * algo + p1)
    centers = _sift_like(rng, 40)
    words = np.clip(centers[rng.integers(0, 40, n_words)].astype(np.int32) + rng.integers(-12, 13, (n_words, 128)), 0, 255).astype(np.uint8)
    words[5] = words[6]
    words[100:110] = words[99]
    proj = rng.standard_normal((64, 128)).astype(np.float32)
    thr = rng.standard_normal((n_words, 64)).astype(np.float32)
    ix = flann_ref.Index.build_forced(words, algo, p1, p2, autotuned_checks=17, seed=n_words)
    path = str(tmp_path / "vocab.bin")
    begin, end = flann_ref.write_reference_vocabulary(path, words, proj, thr, ix, rng, with_entries=True)
    ix.close()
    ref = flann_ref.Index.load(words, path, begin)
    queries = np.concatenate([np.clip(centers[rng.integers(0, 40, 700)].astype(np.int32) + rng.integers(-20, 21, (700, 128)), 0, 255),
                              words[:120], rng.integers(0, 256, (80, 128))]).astype(np.uint8)
    for k in (1, 5):
        for checks in (1, 32, 256, -2):
            rids, rd = ref.knn(queries, k, num_checks=checks, with_dists=True)
            a, ids, dists, ms = _device_search(L, path, queries, k, checks)
            assert a == algo
            assert (ids == rids).all(), (k, checks, int((ids != rids).sum()))
            assert (dists == rd).all()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algo,p1,p2,n_words", [(flann_ref.KDTREE, 1, 0, 2), (flann_ref.KDTREE, 4, 0, 33), (flann_ref.KMEANS, 2, 1, 5),
                                                (flann_ref.KMEANS, 4, 2, 70), (flann_ref.LINEAR, 0, 0, 1), (flann_ref.LINEAR, 0, 0, 7)])
def test_device_flann_search_on_tiny_vocabularies(tmp_path, algo, p1, p2, n_words):
    """Fewer words than neighbours asked for (the rest of the row is kInvalidWordId), trees of one or two levels, zero checks: the
    device search, the host restatement and the reference's FLANN agree."""
    _need_ref()
    L = _host()
    rng = np.random.default_rng(1000 + n_words + algo)
    words = _sift_like(rng, n_words)
    ix = flann_ref.Index.build_forced(words, algo, p1, p2, autotuned_checks=3, seed=n_words)
    path = str(tmp_path / "vocab.bin")
    begin, end = flann_ref.write_reference_vocabulary(path, words, rng.standard_normal((64, 128)).astype(np.float32),
                                                      rng.standard_normal((n_words, 64)).astype(np.float32), ix)
    ix.close()
    ref = flann_ref.Index.load(words, path, begin)
    queries = np.concatenate([_sift_like(rng, 130), words]).astype(np.uint8)
    for k in (1, 5, 8):
        for checks in (0, 1, 256):
            rids, rd = ref.knn(queries, k, num_checks=checks, with_dists=True)
            a, ids, dists, ms = _device_search(L, path, queries, k, checks)
            ha, hids, hd, _ = _product_search(L, path, queries, k, checks, 1)
            assert a == ha == algo
            assert (hids == rids).all() and (hd == rd).all(), ("host", k, checks)
            assert (ids == rids).all() and (dists == rd).all(), ("device", k, checks)
    ref.close()


@pytest.mark.gpu
def test_device_flann_index_upload_is_bounds_checked(dsm):
    """dsm_retrieval_set_flann_index validates every index the kernel would follow: a child that points backwards (a cycle), a
    leaf outside the vocabulary, a pivot outside its array are errors at upload."""
    from dagsfm_amd import capi
    rng = np.random.default_rng(3)
    words = rng.integers(0, 256, (64, 128)).astype(np.uint8)
    dsm.retrieval_set_vocabulary(words, rng.standard_normal((64, 128)).astype(np.float32), rng.standard_normal((64, 64)).astype(np.float32))

    class KdNode(ctypes.Structure):
        _fields_ = [("divfeat", ctypes.c_int32), ("divval", ctypes.c_float), ("child1", ctypes.c_int32), ("child2", ctypes.c_int32)]

    class FlannIndex(ctypes.Structure):
        _fields_ = [("algorithm", ctypes.c_int32), ("num_checks", ctypes.c_int32), ("num_words", ctypes.c_uint32), ("branching", ctypes.c_int32),
                    ("cb_index", ctypes.c_float), ("km_root", ctypes.c_int32), ("n_kd_nodes", ctypes.c_uint32), ("n_kd_roots", ctypes.c_uint32),
                    ("kd_nodes", ctypes.c_void_p), ("kd_roots", ctypes.c_void_p), ("n_km_nodes", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                    ("km_nodes", ctypes.c_void_p), ("n_km_childs", ctypes.c_uint64), ("km_childs", ctypes.c_void_p), ("n_km_points", ctypes.c_uint64),
                    ("km_points", ctypes.c_void_p), ("n_pivot_floats", ctypes.c_uint64), ("pivots", ctypes.c_void_p)]

    def try_upload(nodes):
        arr = (KdNode * len(nodes))(*[KdNode(*n) for n in nodes])
        roots = (ctypes.c_int32 * 1)(0)
        ix = FlannIndex(algorithm=1, num_checks=8, num_words=64, n_kd_nodes=len(nodes), n_kd_roots=1,
                        kd_nodes=ctypes.cast(arr, ctypes.c_void_p), kd_roots=ctypes.cast(roots, ctypes.c_void_p))
        L = dsm._L
        L.dsm_retrieval_set_flann_index.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        return L.dsm_retrieval_set_flann_index(dsm._h, ctypes.byref(ix))

    good = [(5, 100.0, 1, 2), (3, 0.0, -1, -1), (9, 0.0, -1, -1)]
    assert try_upload(good) == 0
    assert try_upload([(5, 100.0, 0, 2), (3, 0.0, -1, -1), (9, 0.0, -1, -1)]) != 0   # a child that is the node itself: a cycle
    assert try_upload([(5, 100.0, 1, 2), (3, 0.0, -1, -1), (64, 0.0, -1, -1)]) != 0  # leaf outside the vocabulary
    assert try_upload([(128, 100.0, 1, 2), (3, 0.0, -1, -1), (9, 0.0, -1, -1)]) != 0  # split dimension outside the descriptor
    assert try_upload([(5, 100.0, 1, 7), (3, 0.0, -1, -1), (9, 0.0, -1, -1)]) != 0   # child outside the node array
    # k-means trees: 64-bit file offsets narrowed to 32-bit device offsets -- an offset near 2^64 must not wrap past the check
    class KmNode(ctypes.Structure):
        _fields_ = [("pivot", ctypes.c_uint64), ("radius", ctypes.c_float), ("variance", ctypes.c_float), ("size", ctypes.c_int32),
                    ("first_child", ctypes.c_uint32), ("num_childs", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("first_point", ctypes.c_uint64)]

    def try_km(pivot0, first_point1, size1=32):
        # a root with two leaves of 32 points each
        nodes = (KmNode * 3)(KmNode(pivot0, 1.0, 1.0, 64, 0, 2, 0, 0), KmNode(128, 1.0, 1.0, size1, 0, 0, 0, first_point1), KmNode(256, 1.0, 1.0, 32, 0, 0, 0, 32))
        childs = (ctypes.c_int32 * 2)(1, 2)
        points = (ctypes.c_uint64 * 64)(*range(64))
        pivots = (ctypes.c_float * 384)()
        ix = FlannIndex(algorithm=2, num_checks=8, num_words=64, branching=2, cb_index=0.2, km_root=0, n_km_nodes=3,
                        km_nodes=ctypes.cast(nodes, ctypes.c_void_p), n_km_childs=2, km_childs=ctypes.cast(childs, ctypes.c_void_p),
                        n_km_points=64, km_points=ctypes.cast(points, ctypes.c_void_p), n_pivot_floats=384, pivots=ctypes.cast(pivots, ctypes.c_void_p))
        L = dsm._L
        L.dsm_retrieval_set_flann_index.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        return L.dsm_retrieval_set_flann_index(dsm._h, ctypes.byref(ix))

    assert try_km(0, 0) == 0
    assert try_km(384, 0) != 0                            # one node past the pivot array
    assert try_km(2 ** 64 - 128, 0) != 0                  # pivot + 128 wraps to 0
    assert try_km(0, 2 ** 64 - 16) != 0                   # first_point + size wraps to 16
    assert try_km(0, 40) != 0                             # 40 + 32 > 64 points
    assert try_km(0, 0, size1=-1) != 0
    L = dsm._L
    assert L.dsm_retrieval_set_flann_index(dsm._h, None) == 0  # back to the exact search
