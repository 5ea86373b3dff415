#!/usr/bin/env python3
"""Writes tests/golden/vocab_flann_{linear,kdtree,kmeans}.bin + vocab_flann_expected.npz: small vocabulary-tree files in the
reference's layout whose middle section is the reference's OWN flann saveIndex output, and what the reference's own
knnSearch over the index LOADED from each file returns for fixed queries (ids and float distances, num_checks 32 and 256,
5 neighbours).  Needs oracle/_ref/libflann_ref.so, i.e. /root/reference (make -C oracle ref); the outputs are committed
so that machines without the reference can still check the product's reader and its FLANN-compatible search.

    python tools/make_flann_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import synthetic  # noqa: E402
from tests import flann_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    assert flann_ref.load() is not None, "build oracle/_ref/libflann_ref.so first (make -C oracle ref; needs /root/reference)"
    rng = np.random.default_rng(11)
    scene = synthetic.Scene(6, 1024, seed=4)
    desc = np.concatenate([scene.image(i)[0] for i in range(6)])
    n_words = 160
    words = desc[rng.choice(len(desc), n_words, replace=False)].copy()
    proj = rng.standard_normal((64, 128)).astype(np.float32)
    thr = rng.standard_normal((n_words, 64)).astype(np.float32)
    queries = np.concatenate([synthetic.Scene(2, 256, seed=21).image(0)[0], words[:16], rng.integers(0, 256, (32, 128)).astype(np.uint8)])
    expected = {"queries": queries, "words": words, "projection": proj, "thresholds": thr}
    for name, algo, p1, p2 in (("linear", flann_ref.LINEAR, 0, 0), ("kdtree", flann_ref.KDTREE, 4, 0), ("kmeans", flann_ref.KMEANS, 8, 3)):
        ix = flann_ref.Index.build_forced(words, algo, p1, p2, autotuned_checks=24, seed=5)
        path = os.path.join(OUT, "vocab_flann_%s.bin" % name)
        begin, end = flann_ref.write_reference_vocabulary(path, words, proj, thr, ix)
        ix.close()
        loaded = flann_ref.Index.load(words, path, begin)  # VisualIndex::Read: the search runs over the LOADED index
        assert loaded.end_offset == end and loaded.algorithm() == algo
        expected[name + "_range"] = np.array([begin, end], np.int64)
        for checks in (32, 256):
            ids, dists = loaded.knn(queries, 5, num_checks=checks, with_dists=True)
            expected["%s_ids_%d" % (name, checks)] = ids
            expected["%s_dists_%d" % (name, checks)] = dists
        loaded.close()
        print(name, "index bytes", end - begin, "file", os.path.getsize(path))
    np.savez_compressed(os.path.join(OUT, "vocab_flann_expected.npz"), **expected)


if __name__ == "__main__":
    main()
