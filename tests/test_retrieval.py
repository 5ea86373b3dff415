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
@pytest.mark.parametrize("n_words", [3, 100, 1000])
def test_device_word_assignment_is_exact(dsm, n_words):
    scene, ims, voc = _scene(3, 300, n_words)
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
