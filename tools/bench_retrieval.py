#!/usr/bin/env python3
"""Candidate-pair generation by vocabulary-tree retrieval on one MI355X: index N images, query
every image, keep (image, retrieved) with image < retrieved -- VocabSimilarityGraph::Run
(/root/reference/src/graph/similarity_graph.cpp:101-199) with its defaults (100 images per query, 5 nearest words).

    python tools/bench_retrieval.py [--images 2000] [--feats 4096] [--words 65536] [--cpu-images 2]

Prints one JSON line: images/s of index + query, device times, the candidate-pair count, the work of the dominant kernel
(word assignment: 2 * 128 * feats * words int8 ops per image on the VALU) and a CPU baseline = oracle/retrieval.cc
(exact nearest words like the device path; the reference itself asks FLANN for approximate ones) timed on a few query
images with the whole index built."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2000)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--words", type=int, default=65536)
    ap.add_argument("--num-images", type=int, default=100)
    ap.add_argument("--neighbors", type=int, default=5)
    ap.add_argument("--cpu-images", type=int, default=0, help="time the CPU oracle's word assignment on this many images (0 = skip)")
    a = ap.parse_args()
    scene = synthetic.Scene(a.images, a.feats, seed=0)
    ims = [scene.image(i) for i in range(a.images)]
    voc = synthetic.vocabulary(scene, a.words, seed=1)
    ctx = capi.Context(0)
    ctx.set_images([im[0] for im in ims])
    ctx.retrieval_set_vocabulary(*voc)
    ctx.retrieval_index()  # warm-up (allocations)
    t0 = time.perf_counter()
    ctx.retrieval_index()
    t1 = time.perf_counter()
    res = ctx.retrieval_query(a.images, a.neighbors, min(a.num_images, a.images))
    t2 = time.perf_counter()
    t_index, t_query = ctx.retrieval_time()
    pairs = set()
    for q, (ids, sc) in enumerate(res):
        for d in ids:
            if q < int(d):
                pairs.add((q, int(d)))
    self_first = sum(1 for q, (ids, sc) in enumerate(res) if len(ids) and ids[0] == q)
    out = {"metric": "images indexed + queried per second (vocabulary-tree candidate pairs)", "value": a.images / (t2 - t0),
           "unit": "images/s", "images": a.images, "feats": a.feats, "words": a.words, "neighbors": a.neighbors,
           "images_per_query": a.num_images, "index_s": t1 - t0, "query_s": t2 - t1, "index_device_ms": t_index,
           "query_device_ms": t_query, "candidate_pairs": len(pairs), "queries_retrieving_themselves_first": self_first,
           "word_assignment_int8_ops_per_image": 2.0 * 128 * a.feats * a.words}
    if a.cpu_images > 0:
        from tests import oracle_lib
        orc = oracle_lib.RetrievalOracle(*voc)
        n = a.cpu_images
        t = time.perf_counter()
        th = [threading.Thread(target=orc.find_word_ids, args=(ims[i][0], a.neighbors)) for i in range(n)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t
        out["cpu_baseline"] = {"value": n / dt, "unit": "images/s (word assignment only)", "cores": n, "kind": "port",
                               "sample": "%d images, exact nearest words of oracle/retrieval.cc, one thread per image, %.1f s" % (n, dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
