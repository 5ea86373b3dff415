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
