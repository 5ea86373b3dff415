#!/usr/bin/env python3
"""The reference-identical visual-word search (VisualIndex::FindWordIds over the vocabulary file's own FLANN index) on the host's
threads and on the device (VERDICT r04 "next" item 6: "a number for the host one", then the device kernel >= 10 x the
all-threads host rate).

Builds a vocabulary file with a REAL FLANN index (the reference's FLANN compiled where it lies, oracle/_ref/libflann_ref.so: kd-trees
x 4 and k-means branching 32 over --words words drawn from the scene's descriptors), then searches the descriptors of --images x
--feats features with num_checks 256:
  host    dagsfm_amd/host/flann_index.cc (the restatement the shim used in round 4), 1 thread on a sample and all threads on the rest
  device  csrc/flann_search.hip through dsm_retrieval_set_flann_index / dsm_retrieval_flann_search, all queries
and compares the ids and distances of the two on every query the host searched.  One JSON line.
    python tools/bench_flann_search.py [--images 500] [--feats 4096] [--words 65536] [--checks 256]"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import synthetic  # noqa: E402
from tests import flann_ref  # noqa: E402
from tests.test_retrieval_flann import _device_search, _host, _product_search  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=500)
    ap.add_argument("--feats", type=int, default=4096)
    ap.add_argument("--words", type=int, default=65536)
    ap.add_argument("--checks", type=int, default=256)
    ap.add_argument("--host-queries", type=int, default=200000, help="queries the all-threads host run searches (a prefix)")
    ap.add_argument("--host-queries-1t", type=int, default=4000)
    a = ap.parse_args()
    if flann_ref.load() is None:
        raise SystemExit("oracle/_ref/libflann_ref.so is missing (make -C oracle ref where /root/reference exists)")
    scene = synthetic.Scene(a.images, a.feats, seed=0)
    desc = np.concatenate([scene.image(i)[0] for i in range(a.images)])
    rng = np.random.default_rng(1)
    words = desc[rng.choice(len(desc), a.words, replace=False)].copy()
    proj = rng.standard_normal((64, 128)).astype(np.float32)
    thr = np.zeros((a.words, 64), np.float32)
    L = _host()
    threads = len(os.sched_getaffinity(0))
    out = {"metric": "visual-word searches per second, FLANN-compatible (reference-identical) search, num_checks %d" % a.checks,
           "queries": int(len(desc)), "words": a.words, "host_threads": threads, "indices": {}}
    tmp = tempfile.mkdtemp()
    for name, algo, p1, p2 in (("kdtree_x4", flann_ref.KDTREE, 4, 0), ("kmeans_b32", flann_ref.KMEANS, 32, 5)):
        t0 = time.perf_counter()
        ix = flann_ref.Index.build_forced(words, algo, p1, p2, autotuned_checks=32, seed=3)
        t_build = time.perf_counter() - t0
        path = os.path.join(tmp, name + ".bin")
        flann_ref.write_reference_vocabulary(path, words, proj, thr, ix)
        ix.close()
        res = {"build_s_reference_flann": t_build}
        for k in (1, 5):  # VisualIndex::Add searches 1 neighbour, ::Query num_nearest_neighbors = 5
            n1, nall = min(a.host_queries_1t, len(desc)), min(a.host_queries, len(desc))
            t0 = time.perf_counter()
            _, ids1, d1, _ = _product_search(L, path, desc[:n1], k, a.checks, 1)
            t1 = time.perf_counter()
            _, idsh, dh, _ = _product_search(L, path, desc[:nall], k, a.checks, threads)
            t2 = time.perf_counter()
            _device_search(L, path, desc, k, a.checks)  # warm-up at full size: context, code objects, the scratch allocations
            t3 = time.perf_counter()
            algo_d, idsd, dd, ms = _device_search(L, path, desc, k, a.checks)
            t4 = time.perf_counter()
            assert algo_d == algo, algo_d
            same = bool((idsd[:nall] == idsh).all() and (dd[:nall] == dh).all() and (ids1 == idsh[:n1]).all())
            # (the timed host calls include reading the vocabulary file and parsing the index: subtract nothing, report the sizes)
            res["k%d" % k] = {"host_1_thread_searches_per_s": n1 / (t1 - t0), "host_1_thread_queries": n1,
                              "host_all_threads_searches_per_s": nall / (t2 - t1), "host_all_threads_queries": nall,
                              "device_kernel_ms": ms, "device_searches_per_s_kernel": len(desc) / (1e-3 * ms),
                              "device_call_s_incl_file_parse_upload_download": t4 - t3,
                              "device_over_host_all_threads": (len(desc) / (1e-3 * ms)) / (nall / (t2 - t1)),
                              "ids_and_distances_identical_on_the_host_sample": same}
            if not same:
                res["k%d" % k]["mismatching_queries"] = int(((idsd[:nall] != idsh).any(axis=1)).sum())
        out["indices"][name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
