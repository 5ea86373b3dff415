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
    ap.add_argument("--flann", default="", choices=["", "kdtree", "kmeans"],
                    help="word_search = kFlann: a FLANN index (kd-trees x 4 / k-means branching 32) built over the vocabulary by the reference's own "
                         "FLANN (oracle/_ref/libflann_ref.so), written into a vocabulary file in the reference's layout, parsed by the host shim and "
                         "searched on the device (csrc/flann_search.hip) with --checks; default: the exact MFMA search")
    ap.add_argument("--checks", type=int, default=256)
    ap.add_argument("--verify", type=int, default=0, help="num_images_after_verification: also time the spatial re-ranking")
    a = ap.parse_args()
    scene = synthetic.Scene(a.images, a.feats, seed=0)
    ims = [scene.image(i) for i in range(a.images)]
    voc = synthetic.vocabulary(scene, a.words, seed=1)
    ctx = capi.Context(0)
    ctx.set_images([im[0] for im in ims])
    ctx.retrieval_set_vocabulary(*voc)
    if a.flann:
        import ctypes
        import tempfile
        from tests import flann_ref
        if flann_ref.load() is None:
            raise SystemExit("oracle/_ref/libflann_ref.so is missing (make -C oracle ref where /root/reference exists)")
        algo, p1, p2 = (flann_ref.KDTREE, 4, 0) if a.flann == "kdtree" else (flann_ref.KMEANS, 32, 5)
        ix = flann_ref.Index.build_forced(voc[0], algo, p1, p2, autotuned_checks=32, seed=3)
        vpath = os.path.join(tempfile.mkdtemp(), "vocab_tree.bin")
        flann_ref.write_reference_vocabulary(vpath, voc[0], voc[1], voc[2], ix)
        ix.close()
        H = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dagsfm_amd", "libdagsfm_host.so"))
        H.dsm_host_flann_attach.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int]
        rc = H.dsm_host_flann_attach(vpath.encode(), ctx._h, a.checks)
        if rc != algo:
            raise SystemExit("dsm_host_flann_attach failed: %d" % rc)
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
           "word_assignment_int8_ops_per_image": 2.0 * 128 * a.feats * a.words,
           "word_search": ("flann %s, num_checks %d (the reference's approximate search on the device: csrc/flann_search.hip)" % (a.flann, a.checks)) if a.flann
                          else "exact (int8 MFMA; a deviation from the reference)"}
    if a.verify > 0:
        # spatial re-ranking (QueryOptions::num_images_after_verification): candidate tuples on the device, then the host
        # shim's SpatialRerank per query (called here one query after the other through its C export; the shim itself
        # spreads the queries over VocabSimilaritySearchOptions::num_threads workers).  Feature shapes that are consistent
        # between the views of a scene point: scale = the point's own x the image's zoom, orientation = the point's + the
        # image's roll.
        import ctypes
        from tests.test_retrieval import _host_lib
        H = _host_lib()
        rng = np.random.default_rng(5)
        base_s, base_o = rng.uniform(1.5, 6.0, scene.n_pool), rng.uniform(-3.0, 3.0, scene.n_pool)
        geoms = []
        for im in ims:
            n = len(im[0])
            pid = np.asarray(im[3])
            zoom, roll = rng.uniform(0.7, 1.4), rng.uniform(-0.4, 0.4)
            sc_ = np.where(pid >= 0, base_s[np.maximum(pid, 0)] * zoom, rng.uniform(1.5, 6.0, n))
            or_ = np.where(pid >= 0, base_o[np.maximum(pid, 0)] + roll, rng.uniform(-3.0, 3.0, n))
            geoms.append(np.c_[im[1][:, 0], im[1][:, 1], sc_, or_].astype(np.float32))
        row0 = np.concatenate([[0], np.cumsum([len(g) for g in geoms])])
        gall = np.ascontiguousarray(np.concatenate(geoms))
        ni = min(a.num_images, a.images)
        t3 = time.perf_counter()
        offs, tup = ctx.retrieval_matches(res, num_neighbors=a.neighbors, max_num_images=ni)
        t4 = time.perf_counter()
        idf = ctx.retrieval_idf(a.words)
        lut = np.array([H.dsm_host_sv_hamming_weight(h) for h in range(65)], np.float32)
        w_all = np.ascontiguousarray((lut[tup[:, 3] & 255] * (idf[tup[:, 3] >> 8] * idf[tup[:, 3] >> 8]).astype(np.float32)).astype(np.float32))
        dbg_all = np.ascontiguousarray(gall[row0[tup[:, 1]] + tup[:, 2]])
        t_host = 0.0
        changed = 0
        for q in range(a.images):
            lo, hi = int(offs[q]), int(offs[q + 1])
            ids = np.ascontiguousarray(res[q][0], np.uint32).copy()
            scq = np.ascontiguousarray(res[q][1], np.float32).copy()
            tq, wq, dq = tup[lo:hi], w_all[lo:hi], dbg_all[lo:hi]
            t5 = time.perf_counter()
            n_after = H.dsm_host_spatial_rerank(len(geoms[q]), geoms[q].ctypes.data, hi - lo, tq.ctypes.data, wq.ctypes.data, dq.ctypes.data,
                                                a.verify, len(ids), ids.ctypes.data, scq.ctypes.data)
            t_host += time.perf_counter() - t5
            changed += list(ids[:n_after]) != list(res[q][0][:n_after])
        out["spatial_reranking"] = {"num_images_after_verification": a.verify, "candidate_tuples": int(offs[-1]),
                                    "device_matches_s": t4 - t3, "host_rerank_s_one_thread": t_host,
                                    "queries_reordered": changed, "images_verified_per_s_one_thread": a.images * ni / max(t_host, 1e-9)}
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
