#!/usr/bin/env python3
"""Differential fuzz of the vocabulary-tree retrieval (dsm_retrieval_set_vocabulary / _index / _query) against
oracle/retrieval.cc on ONE context: many small seeded collections with awkward shapes -- vocabularies of 1 .. 3 000
words with duplicated words (equal distances), images with 0, 1, a few or a few hundred features, duplicated images
(equal scores), words nobody uses, 1 .. 8 neighbours, max_num_images from 1 to more than there are images -- image lists
and scores must be bit-identical, as must the word assignment and the spatially re-ranked lists (device candidate tuples +
the host shim's vote-and-verify against the oracle's verified query).

  python tools/fuzz_retrieval.py [--cases 60] [--seed 1]

Test infrastructure: the oracle is the checker here, as in tests/."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402


def run_fuzz(ctx, n_cases, seed, log=print):
    from tests import oracle_lib
    from tests.test_retrieval import _host_lib, _host_rerank
    H = _host_lib()
    bad = total = 0
    for c in range(n_cases):
        rng = np.random.default_rng([seed, c])
        n_img = int(rng.choice([1, 2, 3, 5, 9, 17, 40]))
        feats = int(rng.choice([8, 60, 200, 500]))
        n_words = int(rng.choice([1, 2, 3, 7, 64, 100, 257, 1000, 3000]))
        k = int(rng.choice([1, 2, 5, 8]))
        max_images = int(rng.choice([1, 2, 5, 100]))
        scene = synthetic.Scene(n_img, feats, seed=int(rng.integers(0, 2**31)), n_pool=max(2 * feats, 16))
        words, proj, thr = synthetic.vocabulary(scene, n_words, seed=int(rng.integers(0, 1000)))
        words = words.copy()
        if n_words > 3 and rng.random() < 0.5:
            words[3::4] = words[2::4][:len(words[3::4])]  # duplicated words: equal distances everywhere
        descs = []
        for i in range(n_img):
            d = scene.image(i)[0]
            r = rng.random()
            if r < 0.15:
                d = d[:0]
            elif r < 0.3:
                d = d[:int(rng.integers(1, 4))]
            elif r < 0.6:
                d = d[:int(rng.integers(1, len(d) + 1))]
            descs.append(np.ascontiguousarray(d))
        if n_img > 2 and rng.random() < 0.4:
            descs[-1] = descs[0].copy()  # a duplicate image: equal scores for every query
        ctx.set_images(descs)
        ctx.retrieval_set_vocabulary(words, proj, thr)
        ctx.retrieval_index()
        res = ctx.retrieval_query(n_img, num_neighbors=k, max_num_images=max_images)
        # keypoint geometry for the spatial re-ranking: random affine shapes
        geoms = []
        for d in descs:
            n = len(d)
            s_, o_ = rng.uniform(1, 6, n), rng.uniform(-3.1, 3.1, n)
            geoms.append(oracle_lib.keypoint_geometry(np.c_[rng.uniform(0, 1000, n), rng.uniform(0, 750, n), s_ * np.cos(o_), -s_ * np.sin(o_),
                                                            s_ * np.sin(o_), s_ * np.cos(o_)].astype(np.float32)))
        orc = oracle_lib.RetrievalOracle(words, proj, thr)
        for i, d in enumerate(descs):
            orc.add_geom(i, d, geoms[i])
        orc.prepare()
        naf = int(rng.choice([1, 2, 5, 100]))
        offs, tup = ctx.retrieval_matches(res, num_neighbors=k, max_num_images=max_images)
        idf = ctx.retrieval_idf(n_words)
        nb = 0
        for q, d in enumerate(descs):
            if len(d):  # the re-ranked list: device tuples + host shim against the oracle's verified query
                ref_ids, ref_sc = orc.query_verified(d, geoms[q], k, max_images, naf)
                got_ids, got_sc = _host_rerank(H, geoms[q], tup[int(offs[q]):int(offs[q + 1])], idf, geoms, naf, res[q][0], res[q][1])
                if list(got_ids) != list(ref_ids) or not (got_sc == ref_sc).all():
                    nb += 1
                    if nb <= 3:
                        log("MISMATCH case %d query %d (re-ranked): %s %s vs %s %s" % (c, q, list(got_ids)[:6], list(got_sc)[:3], list(ref_ids)[:6], list(ref_sc)[:3]))
            ids, sc = orc.query(d, k, max_images)
            if list(res[q][0]) != list(ids) or not (np.asarray(res[q][1]) == sc).all():
                nb += 1
                if nb <= 3:
                    log("MISMATCH case %d query %d: device %s %s, oracle %s %s" % (c, q, list(res[q][0])[:6], list(res[q][1])[:3], list(ids)[:6], list(sc)[:3]))
            if len(d):
                got = ctx.retrieval_debug_word_ids(q, len(d), k)
                if not (got == orc.find_word_ids(d, k)).all():
                    nb += 1
                    if nb <= 3:
                        log("MISMATCH case %d image %d: word ids" % (c, q))
        bad += nb
        total += n_img
        log("case %d: %d images (%s feats), %d words, k %d, max_num_images %d: %d mismatches" %
            (c, n_img, [len(d) for d in descs][:10], n_words, k, max_images, nb))
    return total, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    ctx = capi.Context(0)
    total, bad = run_fuzz(ctx, args.cases, args.seed, log=lambda s: print(s, flush=True))
    print("FUZZ RESULT: %d queries, %d mismatches" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
