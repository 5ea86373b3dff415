"""Vocabulary-tree retrieval = candidate-pair generation.

CPU: oracle/retrieval.cc behaves as the reference's own test expects (visual_index_test.cc:52-106: two images indexed,
querying with the first image's descriptors returns both, the first with the larger score; max_num_images truncates).
GPU: word assignment, the whole index (through the query results) and the retrieval lists are bit-identical to the oracle."""
import numpy as np
import pytest

from dagsfm_amd import capi, synthetic
from tests import oracle_lib


def _random_desc(rng, n):
    return rng.integers(0, 256, (n, 128)).astype(np.uint8)


def test_oracle_behaves_like_reference_visual_index_test():
    """visual_index_test.cc:52-106 with a 100-word vocabulary over random descriptors."""
    rng = np.random.default_rng(0)
    train = _random_desc(rng, 1000)
    words = train[rng.choice(1000, 100, replace=False)]
    q, _ = np.linalg.qr(rng.normal(size=(128, 128)))
    proj = q[:64].astype(np.float32)
    thr = (words.astype(np.float32) @ proj.T).astype(np.float32)
    orc = oracle_lib.RetrievalOracle(words, proj, thr)
    d1, d2 = _random_desc(rng, 50), _random_desc(rng, 50)
    orc.add(1, d1)
    orc.add(2, d2)
    orc.prepare()
    ids, sc = orc.query(d1)
    assert list(ids) == [1, 2] and sc[0] > sc[1]
    ids, sc = orc.query(d1, max_num_images=1)
    assert list(ids) == [1]
    ids, sc = orc.query(d1, max_num_images=3)
    assert list(ids) == [1, 2] and sc[0] > sc[1]
    # exact nearest words: against numpy
    w = orc.find_word_ids(d1, 5)
    dist = ((d1[:, None, :].astype(np.int64) - words[None, :, :].astype(np.int64)) ** 2).sum(-1)
    ref = np.argsort(dist, axis=1, kind="stable")[:, :5]
    assert (w == ref).all()


def _scene(n_img, feats, words):
    scene = synthetic.Scene(n_img, feats, seed=9, n_pool=4 * feats)
    ims = [scene.image(i) for i in range(n_img)]
    voc = synthetic.vocabulary(scene, words, seed=1)
    return scene, ims, voc


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["mfma", "valu"])
@pytest.mark.parametrize("n_words", [3, 100, 1000])
def test_device_word_assignment_is_exact(dsm, n_words, kernel, monkeypatch):
    """Both forms of the word assignment (int8 MFMA tiles with the per-lane top-8 epilogue = the product path; the
    LDS-tiled v_dot4 scan) against the oracle's exact search, for k = 1, 5, 8 neighbours.  Every fourth word is a copy of
    its predecessor, so that equal distances occur everywhere and must resolve to the lower word id."""
    if kernel == "valu":
        monkeypatch.setenv("DSM_VOCAB_ASSIGN_VALU", "1")
    scene, ims, voc = _scene(3, 300, n_words)
    words = voc[0].copy()
    words[3::4] = words[2::4][:len(words[3::4])]
    voc = (words,) + tuple(voc[1:])
    descs = [ims[0][0], ims[1][0][:129], ims[2][0][:1]]
    dsm.set_images(descs)
    dsm.retrieval_set_vocabulary(*voc)
    orc = oracle_lib.RetrievalOracle(*voc)
    for i, d in enumerate(descs):
        for k in (1, 5, 8):
            got = dsm.retrieval_debug_word_ids(i, len(d), k)
            assert (got == orc.find_word_ids(d, k)).all(), (i, k)


@pytest.mark.gpu
@pytest.mark.parametrize("n_img,feats,n_words,max_images", [(12, 400, 256, 100), (40, 256, 2000, 7), (6, 700, 64, 3)])
def test_device_retrieval_equals_oracle(dsm, n_img, feats, n_words, max_images):
    """Index all images, query all images: image lists and scores bit-identical to the oracle (which also fixes the
    order of equal scores)."""
    scene, ims, voc = _scene(n_img, feats, n_words)
    descs = [im[0] for im in ims]
    descs[1] = descs[1][:feats // 2 + 3]  # ragged
    dsm.set_images(descs)
    dsm.retrieval_set_vocabulary(*voc)
    dsm.retrieval_index()
    res = dsm.retrieval_query(n_img, num_neighbors=5, max_num_images=max_images)
    orc = oracle_lib.RetrievalOracle(*voc)
    for i, d in enumerate(descs):
        orc.add(i, d)
    orc.prepare()
    for q, d in enumerate(descs):
        ids, sc = orc.query(d, 5, max_images)
        assert list(res[q][0]) == list(ids), (q, list(res[q][0])[:8], list(ids)[:8])
        assert (res[q][1] == sc).all(), q
        if n_words >= 256:
            assert res[q][0][0] == q  # with a vocabulary that discriminates, an image retrieves itself first
    t_index, t_query = dsm.retrieval_time()
    assert t_index > 0 and t_query > 0


@pytest.mark.gpu
def test_host_vocab_similarity_graph_over_database(tmp_path, dsm):
    """dagsfm_amd::VocabSimilarityGraph (the C++ mirror of DAGSfM::VocabSimilarityGraph::Run) over a database.db and a
    vocabulary file == the pairs composed from the C-ABI results: (image_id, retrieved) with image_id < retrieved,
    score * 1e3, image ids in ascending order (similarity_graph.cpp:183-194)."""
    import ctypes
    import os
    from tests import dbutil
    n_img = 9
    scene, ims, voc = _scene(n_img, 300, 512)
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims])
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dagsfm_amd", "libdagsfm_host.so"))
    L.dsm_host_write_vocabulary.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.dsm_host_vocab_candidate_pairs.restype = ctypes.c_int64
    L.dsm_host_vocab_candidate_pairs.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_uint64]
    vpath = str(tmp_path / "vocab.bin")
    w, p, t = [np.ascontiguousarray(x) for x in voc]
    assert L.dsm_host_write_vocabulary(vpath.encode(), len(w), w.ctypes.data, p.ctypes.data, t.ctypes.data) == 0
    pairs = np.zeros((1000, 2), np.uint32)
    scores = np.zeros(1000, np.float32)
    n = L.dsm_host_vocab_candidate_pairs(path.encode(), vpath.encode(), 4, 5, pairs.ctypes.data, scores.ctypes.data, 1000)
    assert n > 0
    dsm.set_images([im[0] for im in ims])
    dsm.retrieval_set_vocabulary(*voc)
    dsm.retrieval_index()
    res = dsm.retrieval_query(n_img, 5, 4)
    exp_pairs, exp_scores = [], []
    for q, (ids, sc) in enumerate(res):
        for d, s in zip(ids, sc):
            if q < int(d):
                exp_pairs.append((q + 1, int(d) + 1))  # dbutil's image ids are 1-based
                exp_scores.append(np.float32(s) * np.float32(1e3))
    assert n == len(exp_pairs)
    assert [tuple(x) for x in pairs[:n]] == exp_pairs
    assert (scores[:n] == np.array(exp_scores, np.float32)).all()


def _write_reference_vocabulary(path, words, projection, thresholds, rng, with_entries=False):
    """A vocabulary-tree file in the reference's own layout (VisualIndex<>::Write, retrieval/visual_index.h:586-614):
    words, an opaque FLANN blob (random bytes here -- the reader must find the inverted index without understanding it),
    the inverted index (inverted_index.h:383-420 / inverted_file.h:394-411)."""
    import struct
    w = np.ascontiguousarray(words, np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", w.shape[0], 128))
        f.write(w.tobytes())
        blob = rng.integers(0, 256, int(rng.integers(1000, 5000)), dtype=np.uint8).tobytes()
        # plant the index header's byte pattern inside the blob: the reader must not be fooled by it
        blob = blob[:100] + struct.pack("<ii", w.shape[0], 64) + blob[100:]
        f.write(blob)
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


@pytest.mark.parametrize("with_entries", [False, True])
def test_reads_the_references_vocabulary_file_layout(tmp_path, with_entries):
    """VERDICT r02 (missing 1): the reference's vocabulary file is read as it is (the FLANN blob skipped), no converter."""
    import ctypes
    import os
    rng = np.random.default_rng(3)
    words, proj, thr = _scene(2, 64, 96)[2]
    path = str(tmp_path / "vocab_tree.bin")
    _write_reference_vocabulary(path, words, proj, thr, rng, with_entries)
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dagsfm_amd", "libdagsfm_host.so"))
    L.dsm_host_read_vocabulary.restype = ctypes.c_uint32
    L.dsm_host_read_vocabulary.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    w2 = np.zeros_like(np.ascontiguousarray(words, np.uint8))
    p2 = np.zeros((64, 128), np.float32)
    t2 = np.zeros((len(words), 64), np.float32)
    assert L.dsm_host_read_vocabulary(path.encode(), w2.ctypes.data, p2.ctypes.data, t2.ctypes.data, len(words)) == len(words)
    assert (w2 == words).all() and (p2 == proj).all() and (t2 == thr).all()
    # a truncated file is refused, not misread
    open(path, "r+b").truncate(os.path.getsize(path) - 3)
    assert L.dsm_host_read_vocabulary(path.encode(), None, None, None, 0) == 0


@pytest.mark.gpu
def test_host_vocab_similarity_graph_max_num_features(tmp_path, dsm):
    """VocabTreeMatching.max_num_features (similarity_graph.cpp:77-79, 137-141): every image is indexed and queried with
    its max_num_features features of largest scale only, in ExtractTopScaleFeatures' order; the vocabulary comes from a
    file in the reference's layout."""
    import ctypes
    import os
    import sqlite3
    from tests import dbutil
    n_img, keep = 8, 120
    scene, ims, voc = _scene(n_img, 300, 512)
    rng = np.random.default_rng(8)
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims])
    # distinct keypoint scales (a11 = a22 = s, a12 = a21 = 0 -> ComputeScale = s): the top-scale order is then unique
    con = sqlite3.connect(path)
    orders = []
    for i, im in enumerate(ims):
        n = len(im[0])
        s = rng.permutation(n).astype(np.float32) + 1.0
        k = np.zeros((n, 6), np.float32)
        k[:, :2] = im[1]
        k[:, 2] = s
        k[:, 5] = s
        con.execute("UPDATE keypoints SET data = ? WHERE image_id = ?", (k.tobytes(), i + 1))
        orders.append(np.argsort(-s)[:keep])
    con.commit()
    con.close()
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dagsfm_amd", "libdagsfm_host.so"))
    L.dsm_host_vocab_candidate_pairs2.restype = ctypes.c_int64
    L.dsm_host_vocab_candidate_pairs2.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_uint64]
    vpath = str(tmp_path / "vocab_tree.bin")
    _write_reference_vocabulary(vpath, *voc, rng)
    pairs = np.zeros((1000, 2), np.uint32)
    scores = np.zeros(1000, np.float32)
    n = L.dsm_host_vocab_candidate_pairs2(path.encode(), vpath.encode(), 4, 5, keep, pairs.ctypes.data, scores.ctypes.data, 1000)
    assert n > 0
    dsm.set_images([im[0][o] for im, o in zip(ims, orders)])
    dsm.retrieval_set_vocabulary(*voc)
    dsm.retrieval_index()
    res = dsm.retrieval_query(n_img, 5, 4)
    exp_pairs, exp_scores = [], []
    for q, (ids, sc) in enumerate(res):
        for d, s in zip(ids, sc):
            if q < int(d):
                exp_pairs.append((q + 1, int(d) + 1))
                exp_scores.append(np.float32(s) * np.float32(1e3))
    assert n == len(exp_pairs) and [tuple(x) for x in pairs[:n]] == exp_pairs
    assert (scores[:n] == np.array(exp_scores, np.float32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("delta", [-3, 5])
def test_vocab_similarity_graph_rejects_keypoint_descriptor_mismatch(tmp_path, delta):
    """a database whose keypoints row has more (heap over-read) or fewer (features silently dropped) rows than
    its descriptors row -- the reference CHECK_EQs the two in ExtractTopScaleFeatures (src/feature/utils.cc:84); the shim
    reports an error, with max_num_features and with the spatial re-ranking's geometries alike."""
    import ctypes
    import os
    import sqlite3
    from tests import dbutil
    scene, ims, voc = _scene(4, 200, 64)
    rng = np.random.default_rng(3)
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(im[0], im[1]) for im in ims])
    n = len(ims[2][0]) + delta
    con = sqlite3.connect(path)
    con.execute("UPDATE keypoints SET rows = ?, data = ? WHERE image_id = 3", (n, np.ones((n, 6), np.float32).tobytes()))
    con.commit()
    con.close()
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dagsfm_amd", "libdagsfm_host.so"))
    L.dsm_host_vocab_candidate_pairs2.restype = ctypes.c_int64
    L.dsm_host_vocab_candidate_pairs2.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_uint64]
    vpath = str(tmp_path / "vocab_tree.bin")
    _write_reference_vocabulary(vpath, *voc, rng)
    pairs = np.zeros((100, 2), np.uint32)
    scores = np.zeros(100, np.float32)
    assert L.dsm_host_vocab_candidate_pairs2(path.encode(), vpath.encode(), 3, 5, 50, pairs.ctypes.data, scores.ctypes.data, 100) < 0
    # without max_num_features and without re-ranking the keypoints are never read: the run goes through
    assert L.dsm_host_vocab_candidate_pairs2(path.encode(), vpath.encode(), 3, 5, 0, pairs.ctypes.data, scores.ctypes.data, 100) > 0


# ------------------------------------------------------------------------------------------- spatial re-ranking
# QueryOptions::num_images_after_verification > 0: VisualIndex::Query with geometries (visual_index.h:259-500) +
# VoteAndVerify (vote_and_verify.cc).  Oracle: oracle/spatial_verification.h + oracle_retrieval_query_verified; product:
# dsm_retrieval_matches on the device, dagsfm_amd/host/spatial_verification.cc on the host.
def _host_lib():
    import ctypes
    import os
    H = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dagsfm_amd", "libdagsfm_host.so"))
    H.dsm_host_sv_vote_and_verify.restype = ctypes.c_int
    H.dsm_host_sv_vote_and_verify.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    H.dsm_host_sv_estimate_affine.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    H.dsm_host_sv_keypoint_geometry.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    H.dsm_host_sv_hamming_weight.restype = ctypes.c_float
    H.dsm_host_sv_hamming_weight.argtypes = [ctypes.c_uint32]
    H.dsm_host_spatial_rerank.restype = ctypes.c_uint32
    H.dsm_host_spatial_rerank.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    H.dsm_host_vocab_candidate_pairs3.restype = ctypes.c_int64
    H.dsm_host_vocab_candidate_pairs3.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    return H


def _host_rerank(H, qgeom, tup, idf, geoms, naf, ids, scores):
    lut = np.array([H.dsm_host_sv_hamming_weight(h) for h in range(65)], np.float32)
    w = np.ascontiguousarray((lut[tup[:, 3] & 255] * (idf[tup[:, 3] >> 8] * idf[tup[:, 3] >> 8]).astype(np.float32)).astype(np.float32))
    dbg = np.array([geoms[int(t[1])][int(t[2])] for t in tup], np.float32).reshape(-1, 4)
    ids = np.ascontiguousarray(ids, np.uint32).copy()
    sc = np.ascontiguousarray(scores, np.float32).copy()
    qg = np.ascontiguousarray(qgeom, np.float32)
    tup = np.ascontiguousarray(tup, np.uint32)
    n = H.dsm_host_spatial_rerank(len(qg), qg.ctypes.data, len(tup), tup.ctypes.data, w.ctypes.data, dbg.ctypes.data, naf, len(ids),
                                  ids.ctypes.data, sc.ctypes.data)
    return ids[:n], sc[:n]


def test_spatial_leaves_like_the_reference_tests():
    """retrieval/geometry_test.cc:43-135 (TransformFromMatch: identity, translation, scale, orientation) and
    estimators/affine_transform_test.cc:40-67 on the oracle; then the product's host implementation against the oracle,
    bit for bit: the least-squares affine map (degenerate inputs included), FeatureKeypoint's scale / orientation, and
    VoteAndVerify over random similarity scenes with outliers, zero scales (inf / NaN paths) and single positions."""
    for x in range(3):
        for y in range(3):
            for s in range(1, 5):
                for o in range(3):
                    assert np.allclose(oracle_lib.sv_transform_from_match([x, y, s, o], [x, y, s, o]), [1, 0, 0, 0], atol=1e-6)
            assert np.allclose(oracle_lib.sv_transform_from_match([0, 0, 1, 0], [x, y, 1, 0]), [1, 0, x, y], atol=1e-6)
    for s in range(1, 5):
        assert np.allclose(oracle_lib.sv_transform_from_match([0, 0, 1, 0], [0, 0, s, 0]), [s, 0, 0, 0])
    for o in range(3):
        assert np.allclose(oracle_lib.sv_transform_from_match([0, 0, 1, 0], [0, 0, 1, o]), [1, o, 0, 0])
    for x in np.arange(0, 1, 0.1):
        A = np.array([[x, 0.2, 0.3], [30, 0.2, 0.1]])
        src = np.array([[x, 0], [1, 0], [2, 1]], float)
        dst = (A @ np.c_[src, np.ones(3)].T).T
        Ae = oracle_lib.sv_estimate_affine(src, dst)
        assert (((dst - (Ae @ np.c_[src, np.ones(3)].T).T) ** 2).sum(1) < 1e-6).all()
    H = _host_lib()
    rng = np.random.default_rng(0)
    for n in (3, 4, 5, 9, 10, 40, 300):
        for rep in range(4):
            x1 = np.ascontiguousarray(rng.uniform(0, 1000, (n, 2)))
            x2 = (rng.normal(size=(2, 3)) * [1, 1, 50] @ np.c_[x1, np.ones(n)].T).T + rng.normal(scale=rng.choice([0, 1.0]), size=(n, 2))
            if rep == 3:
                x2[:] = x2[0]
            x2 = np.ascontiguousarray(x2)
            b = np.zeros(6)
            H.dsm_host_sv_estimate_affine(x1.ctypes.data, x2.ctypes.data, n, b.ctypes.data)
            a = oracle_lib.sv_estimate_affine(x1, x2).ravel()
            assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (n, rep)
    kp = np.c_[rng.uniform(0, 100, (50, 2)), rng.normal(size=(50, 4))].astype(np.float32)
    g = np.zeros((50, 4), np.float32)
    H.dsm_host_sv_keypoint_geometry(kp.ctypes.data, 50, g.ctypes.data)
    assert (g == oracle_lib.keypoint_geometry(kp)).all()
    n_pos = 0
    for trial in range(150):
        n = int(rng.choice([0, 2, 3, 5, 20, 100, 400]))
        g1 = np.c_[rng.uniform(0, 1000, n), rng.uniform(0, 750, n), rng.uniform(0.5, 8, n), rng.uniform(-3.2, 3.2, n)].astype(np.float32)
        ang, sc = rng.uniform(-3, 3), rng.uniform(0.3, 3)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        xy2 = (sc * (R @ g1[:, :2].T)).T + rng.uniform(-300, 300, 2) + rng.normal(scale=rng.choice([0, 2, 20]), size=(n, 2))
        g2 = np.c_[xy2, g1[:, 2] * sc * rng.uniform(0.8, 1.25, n), g1[:, 3] + ang + rng.normal(scale=0.1, size=n)].astype(np.float32)
        k = int(rng.uniform(0, 1) * n)
        if k:
            g2[rng.choice(n, k, replace=False), :2] = rng.uniform(0, 1000, (k, 2))
        if trial % 17 == 0 and n:
            g2[:, 2] = 0
        if trial % 19 == 0 and n:
            g1[:, :2] = g1[0, :2]
        a = oracle_lib.sv_vote_and_verify(g1, g2)
        g1c, g2c = np.ascontiguousarray(g1), np.ascontiguousarray(g2)
        assert a == H.dsm_host_sv_vote_and_verify(n, g1c.ctypes.data, g2c.ctypes.data), trial
        n_pos += a > 0
    assert n_pos > 40
    # special values anywhere (zero / huge / infinite / NaN coordinates, scales, orientations): NaN transformations pass the
    # range tests of the vote and arrive at the bin index as INT_MIN coordinates; both sides wrap like x86 does
    specials = np.array([0.0, -0.0, 1e-30, 1e30, np.inf, -np.inf, np.nan, 1.0, -1.0], np.float32)
    for trial in range(400):
        n = int(rng.integers(0, 40))
        g1 = (rng.uniform(0, 1000, (n, 4))).astype(np.float32)
        g2 = (rng.uniform(0, 1000, (n, 4))).astype(np.float32)
        for g in (g1, g2):
            mask = rng.random((n, 4)) < 0.2
            g[mask] = specials[rng.integers(0, len(specials), int(mask.sum()))]
        same = rng.random(n) < 0.3
        g2[same] = g1[same]
        g1c, g2c = np.ascontiguousarray(g1), np.ascontiguousarray(g2)
        assert oracle_lib.sv_vote_and_verify(g1, g2) == H.dsm_host_sv_vote_and_verify(n, g1c.ctypes.data, g2c.ctypes.data), trial
    # a consistent similarity between 200 features is found whole; random positions are not
    n = 200
    g1 = np.c_[rng.uniform(0, 1000, n), rng.uniform(0, 750, n), rng.uniform(1, 5, n), rng.uniform(-3, 3, n)].astype(np.float32)
    R = np.array([[np.cos(0.3), -np.sin(0.3)], [np.sin(0.3), np.cos(0.3)]])
    g2 = np.c_[(1.4 * (R @ g1[:, :2].T)).T + [50, -20], g1[:, 2] * 1.4, g1[:, 3] + 0.3].astype(np.float32)
    assert oracle_lib.sv_vote_and_verify(g1, g2) > 150
    g2[:, :2] = rng.uniform(0, 1000, (n, 2))
    assert oracle_lib.sv_vote_and_verify(g1, g2) < 30
    assert oracle_lib.sv_vote_and_verify(g1[:2], g2[:2]) == 0


def _spatial_case(rng, n_img, feats, n_words):
    scene = synthetic.Scene(n_img, feats, seed=int(rng.integers(0, 2**31)), n_pool=int(feats * 1.5))
    voc = synthetic.vocabulary(scene, n_words, seed=3)
    ims = [scene.image(i) for i in range(n_img)]
    descs = [im[0] for im in ims]
    kps = []
    for im in ims:  # keypoints with an affine shape (6 columns), like the database holds for SIFT features
        n = len(im[0])
        s, o = rng.uniform(1, 6, n), rng.uniform(-3.1, 3.1, n)
        kps.append(np.c_[im[1][:, 0], im[1][:, 1], s * np.cos(o), -s * np.sin(o), s * np.sin(o), s * np.cos(o)].astype(np.float32))
    if n_img > 3:
        descs[2], kps[2] = descs[2][:0], kps[2][:0]
    return voc, descs, kps


def test_host_spatial_rerank_equals_oracle():
    """The host half of the product (1-to-1 assignment with its own bookkeeping, VoteAndVerify, re-ranking) against
    oracle_retrieval_query_verified, with the device's candidate tuples restated in numpy (tests/retrieval_emulation.py):
    image lists and scores bit-identical, for 1 / 3 / 5 neighbours, short and long retrieval lists, an empty image."""
    from tests import retrieval_emulation
    H = _host_lib()
    rng = np.random.default_rng(1)
    total = changed = 0
    for case in range(8):
        n_img, feats, n_words = int(rng.choice([3, 6, 10])), int(rng.choice([40, 120])), int(rng.choice([8, 64, 300]))
        k, max_images, naf = int(rng.choice([1, 3, 5])), int(rng.choice([2, 5, 100])), int(rng.choice([1, 3, 100]))
        voc, descs, kps = _spatial_case(rng, n_img, feats, n_words)
        geoms = [oracle_lib.keypoint_geometry(k_) for k_ in kps]
        orc = oracle_lib.RetrievalOracle(*voc)
        for i, d in enumerate(descs):
            orc.add_geom(i, d, geoms[i])
        orc.prepare()
        tuples, idf = retrieval_emulation.emulate(voc[0], voc[1], voc[2], descs, k, orc)
        for q in range(n_img):
            ids0, sc0 = orc.query(descs[q], k, max_images)
            ref_ids, ref_sc = orc.query_verified(descs[q], geoms[q], k, max_images, naf)
            if len(descs[q]) == 0:
                assert len(ref_ids) == 0
                continue
            got_ids, got_sc = _host_rerank(H, geoms[q], tuples(q, ids0), idf, geoms, naf, ids0, sc0)
            assert list(got_ids) == list(ref_ids) and (got_sc == ref_sc).all(), (case, q)
            assert len(ref_ids) == min(len(ids0), naf)
            total += 1
            changed += list(ref_ids) != list(ids0[:len(ref_ids)])
    assert total > 30 and changed > 5


@pytest.mark.gpu
def test_device_retrieval_matches_and_reranked_database_run(tmp_path, dsm):
    """(1) dsm_retrieval_matches / dsm_get_retrieval_matches / dsm_get_retrieval_idf against the numpy restatement:
    offsets, the tuples in their order, the IDF weights.  (2) VocabSimilarityGraph::Run over a database.db with
    num_images_after_verification > 0 (host shim + device) against the oracle's verified queries: pairs and scores."""
    import ctypes
    from tests import dbutil, retrieval_emulation
    rng = np.random.default_rng(2)
    n_img, feats, n_words, k, max_images, naf = 9, 200, 128, 5, 6, 4
    voc, descs, kps = _spatial_case(rng, n_img, feats, n_words)
    geoms = [oracle_lib.keypoint_geometry(k_) for k_ in kps]
    orc = oracle_lib.RetrievalOracle(*voc)
    for i, d in enumerate(descs):
        orc.add_geom(i, d, geoms[i])
    orc.prepare()
    dsm.set_images(descs)
    dsm.retrieval_set_vocabulary(*voc)
    dsm.retrieval_index()
    res = dsm.retrieval_query(n_img, num_neighbors=k, max_num_images=max_images)
    offs, tup = dsm.retrieval_matches(res, num_neighbors=k, max_num_images=max_images)
    tuples, idf = retrieval_emulation.emulate(voc[0], voc[1], voc[2], descs, k, orc)
    assert (dsm.retrieval_idf(n_words) == idf).all()
    for q in range(n_img):
        exp = tuples(q, res[q][0])
        got = tup[int(offs[q]):int(offs[q + 1])]
        assert got.shape == exp.shape and (got == exp).all(), q
    assert offs[-1] > 500
    # (2) through the database
    path = str(tmp_path / "database.db")
    dbutil.create(path, [(d, kp) for d, kp in zip(descs, kps)], prior=True, kp_cols=6)
    vpath = str(tmp_path / "vocab.bin")
    H = _host_lib()
    H.dsm_host_write_vocabulary.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    w, p, t = [np.ascontiguousarray(a) for a in voc]
    assert H.dsm_host_write_vocabulary(vpath.encode(), n_words, w.ctypes.data, p.ctypes.data, t.ctypes.data) == 0
    pairs = np.zeros((1000, 2), np.uint32)
    scores = np.zeros(1000, np.float32)
    n = H.dsm_host_vocab_candidate_pairs3(path.encode(), vpath.encode(), max_images, k, -1, naf, pairs.ctypes.data, scores.ctypes.data, 1000)
    exp_pairs, exp_scores = [], []
    for q in range(n_img):
        ids, sc = orc.query_verified(descs[q], geoms[q], k, max_images, naf)
        for i, s in zip(ids, sc):
            if q < int(i):
                exp_pairs.append((q + 1, int(i) + 1))
                exp_scores.append(np.float32(s) * np.float32(1e3))
    assert n == len(exp_pairs) and n > 5
    assert [tuple(x) for x in pairs[:n]] == exp_pairs
    assert (scores[:n] == np.array(exp_scores, np.float32)).all()


def test_bin_order_of_the_platform():
    """VoteAndVerify verifies the 30 best-scored voting bins; which bins those are among equal scores, and in which order,
    is in the reference whatever std::unordered_map's iteration and std::partial_sort make of it.  Oracle and product use
    the same containers with the same insertions (this toolchain's libstdc++); an order of our own -- equal scores by
    ascending bin index -- would change about one result in a hundred on random scenes, usually by one or two effective
    inliers.  This test keeps that figure honest."""
    rng = np.random.default_rng(0)
    n_cases = n_diff = 0
    for trial in range(500):
        n = int(rng.choice([5, 20, 60, 150, 400]))
        g1 = np.c_[rng.uniform(0, 1000, n), rng.uniform(0, 750, n), rng.uniform(0.5, 8, n), rng.uniform(-3.2, 3.2, n)].astype(np.float32)
        ang, sc = rng.uniform(-3, 3), rng.uniform(0.3, 3)
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        xy2 = (sc * (R @ g1[:, :2].T)).T + rng.uniform(-300, 300, 2) + rng.normal(scale=rng.choice([0, 2, 20]), size=(n, 2))
        g2 = np.c_[xy2, g1[:, 2] * sc * rng.uniform(0.8, 1.25, n), g1[:, 3] + ang + rng.normal(scale=0.1, size=n)].astype(np.float32)
        k = int(rng.choice([0, 0.3, 0.6, 0.9, 1.0]) * n)
        if k:
            g2[rng.choice(n, k, replace=False), :2] = rng.uniform(0, 1000, (k, 2))
        n_cases += 1
        n_diff += oracle_lib.sv_vote_and_verify(g1, g2, True) != oracle_lib.sv_vote_and_verify(g1, g2, False)
    assert n_diff <= 0.05 * n_cases


def test_hamming_weights_equal_the_references_own_header():
    """HammingDistWeightFunctor<64> (retrieval/utils.h:47-78, a plain standard C++ header compiled where it lies into
    oracle/_ref/libmisc_ref.so): the oracle's table and the host shim's weights are the same 65 floats, and the cut-off is
    the 24 the kernels use."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmisc_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libmisc_ref.so is built only where /root/reference exists (make -C oracle ref)")
    R = ctypes.CDLL(path)
    R.ref_hamming_weight.restype = ctypes.c_float
    R.ref_hamming_weight.argtypes = [ctypes.c_uint32]
    R.ref_max_hamming_distance.restype = ctypes.c_uint32
    L = oracle_lib.load().lib
    L.oracle_retrieval_hamming_weight.restype = ctypes.c_float
    L.oracle_retrieval_hamming_weight.argtypes = [ctypes.c_uint32]
    H = _host_lib()
    assert R.ref_max_hamming_distance() == 24
    for h in range(65):
        ref = np.float32(R.ref_hamming_weight(h))
        assert np.float32(L.oracle_retrieval_hamming_weight(h)) == ref and np.float32(H.dsm_host_sv_hamming_weight(h)) == ref, h
        assert (ref > 0) == (h <= 24)
