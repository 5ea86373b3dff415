#!/usr/bin/env python3
"""How far is the product's default word search (EXACT nearest visual words, on the device) from what the reference
returns (FLANN's approximate search over the index stored in the vocabulary file, num_checks = 256)?  CPU only: the
oracle (oracle/retrieval.cc, exact search) against the same oracle with the reference's own FLANN plugged into every
word search (oracle/_ref/libflann_ref.so, compiled from /root/reference/lib/FLANN).  The vocabulary is built the way
VisualIndex::Build builds it -- flann::hierarchicalClustering (Quantize, visual_index.h:624-665) over training descriptors,
then an index over the words (:517-521): the autotuner's own choice on this machine, and the two tree types it can choose.

    python tools/flann_agreement.py [--words 8192] [--images 40] [--feats 1024] > profiles/r04_flann_agreement.json

Reported per index: share of features whose nearest word agrees (k = 1, what VisualIndex::Add stores), mean overlap of the 5
nearest words (what a query scores), and the Jaccard overlap of the candidate-pair lists VocabSimilarityGraph::Run would
emit (num_images per query).  The product's word_search = flann mode returns the reference's ids themselves
(tests/test_retrieval_flann.py), so this gap is a property of the DEFAULT mode only."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import synthetic  # noqa: E402
from tests import flann_ref, oracle_lib  # noqa: E402


def candidate_pairs(orc, ims, k, num_images):
    pairs = set()
    for q, im in enumerate(ims):
        ids, _ = orc.query(im[0], k, num_images)
        for d in ids:
            if q < int(d):
                pairs.add((q, int(d)))
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--words", type=int, default=8192)
    ap.add_argument("--images", type=int, default=40)
    ap.add_argument("--feats", type=int, default=1024)
    ap.add_argument("--num-images", type=int, default=10, help="VocabSimilaritySearchOptions::num_images (retrieved per query)")
    ap.add_argument("--checks", type=int, default=256)
    a = ap.parse_args()
    assert flann_ref.load() is not None, "make -C oracle ref (needs /root/reference)"
    cache = "/tmp/flann_agreement_words_%d.npy" % a.words
    t0 = time.time()
    if os.path.exists(cache):
        words = np.load(cache)
    else:  # VisualIndex::Quantize over training descriptors of another scene
        n_train = max(40, a.words * 20 // 4096)
        train = synthetic.Scene(n_train, 4096, seed=2)
        d = np.concatenate([train.image(i)[0] for i in range(n_train)])
        w = np.zeros((a.words, 128), np.uint8)
        flann_ref.load().flann_ref_seed(1)
        nc = flann_ref.load().flann_ref_quantize(d.ctypes.data, len(d), a.words, 256, 11, w.ctypes.data)
        words = w[:nc].copy()
        np.save(cache, words)
    rng = np.random.default_rng(0)
    proj = rng.standard_normal((64, 128)).astype(np.float32)
    thr = (proj @ words.astype(np.float32).T).T.astype(np.float32)
    scene = synthetic.Scene(a.images, a.feats, seed=5)
    ims = [scene.image(i) for i in range(a.images)]
    alld = np.concatenate([im[0] for im in ims])
    out = {"words": int(len(words)), "images": a.images, "feats": a.feats, "num_checks": a.checks, "num_neighbors": 5, "num_images": a.num_images,
           "vocabulary": "flann::hierarchicalClustering (branching 256, 11 iterations, k-means++) over synthetic SIFT-like descriptors of a "
                         "training scene; queries from another scene (dagsfm_amd/synthetic.py: unstructured 128-D descriptors -- the hardest "
                         "case for a tree search; real SIFT clusters far better)", "indices": {}}
    exact = oracle_lib.RetrievalOracle(words, proj, thr)
    e1 = exact.find_word_ids(alld, 1)
    e5 = exact.find_word_ids(alld, 5)
    for i, im in enumerate(ims):
        exact.add(i, im[0])
    exact.prepare()
    pairs_exact = candidate_pairs(exact, ims, 5, a.num_images)
    out["exact_seconds"] = time.time() - t0
    cases = [("autotuned (target_precision 0.95, the tuner's own choice here)", None), ("kd-trees x4", (flann_ref.KDTREE, 4, 0)),
             ("kd-trees x16", (flann_ref.KDTREE, 16, 0)), ("k-means 32 x 5 iterations", (flann_ref.KMEANS, 32, 5))]
    for name, forced in cases:
        flann_ref.load().flann_ref_seed(7)
        ix = flann_ref.Index.build(words, 0.95) if forced is None else flann_ref.Index.build_forced(words, *forced)
        f1 = ix.knn(alld, 1, num_checks=a.checks)
        f5 = ix.knn(alld, 5, num_checks=a.checks)
        orc = oracle_lib.RetrievalOracle(words, proj, thr)
        orc.use_flann(ix, a.checks)
        for i, im in enumerate(ims):
            orc.add(i, im[0])
        orc.prepare()
        pairs_flann = candidate_pairs(orc, ims, 5, a.num_images)
        out["indices"][name] = {
            "algorithm": ix.algorithm(),
            "top1_agreement": float((f1[:, 0] == e1[:, 0]).mean()),
            "top5_mean_overlap": float(np.mean([len(set(x) & set(y)) / 5.0 for x, y in zip(f5, e5)])),
            "candidate_pairs_exact": len(pairs_exact), "candidate_pairs_flann": len(pairs_flann),
            "candidate_pairs_jaccard": len(pairs_exact & pairs_flann) / max(1, len(pairs_exact | pairs_flann)),
        }
        ix.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
