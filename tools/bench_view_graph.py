#!/usr/bin/env python3
"""View-graph rotation-cycle filter (ViewGraph::FilterViewGraphCyclesByRotation,
/root/reference/src/graph/view_graph.cpp:115-165) on one MI355X: a kNN-style view graph of N images with K neighbours each
(BASELINE configs[3] shape: 10 000 images, ~200 neighbours), relative rotations of random absolute orientations with a
little noise and a fraction of corrupted edges.

    python tools/bench_view_graph.py [--images 10000] [--neighbors 200] [--corrupt 0.02] [--cpu-edges 20000]

Prints one JSON line: edges/s and triplets/s of dsm_view_graph_filter_cycles (host buffers in, keep flags out: the call
includes the CSR build on the host and both PCIe copies), the kept fraction, and a CPU baseline = oracle/view_graph.cc on
a sub-graph of the first images whose edge count is about --cpu-edges (the device result for that sub-graph is compared
with the oracle's)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi  # noqa: E402


def qmul(a, b):
    w1, x1, y1, z1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    w2, x2, y2, z2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=1)


def build(n_img, k, corrupt, noise, seed):
    rng = np.random.default_rng(seed)
    # neighbours in a window around the image (what a retrieval graph over an ordered capture looks like): many triangles
    half = k // 2
    i = np.repeat(np.arange(n_img), half)
    j = i + np.tile(np.arange(1, half + 1), n_img)
    ok = j < n_img
    pairs = np.stack([i[ok], j[ok]], axis=1).astype(np.uint32)
    absq = rng.normal(size=(n_img, 4))
    absq /= np.linalg.norm(absq, axis=1, keepdims=True)
    conj = absq * np.array([1.0, -1.0, -1.0, -1.0])
    q = qmul(absq[pairs[:, 1]], conj[pairs[:, 0]])
    q += rng.normal(scale=noise, size=q.shape)
    bad = rng.random(len(q)) < corrupt
    rq = rng.normal(size=(int(bad.sum()), 4))
    q[bad] = rq
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return pairs, q, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=10000)
    ap.add_argument("--neighbors", type=int, default=200)
    ap.add_argument("--corrupt", type=float, default=0.02)
    ap.add_argument("--noise", type=float, default=0.002)
    ap.add_argument("--cpu-edges", type=int, default=20000)
    a = ap.parse_args()
    pairs, q, bad = build(a.images, a.neighbors, a.corrupt, a.noise, 0)
    ctx = capi.Context(0)
    keep, nt = ctx.view_graph_filter_cycles(pairs, q)  # warm-up (allocations)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        keep, nt = ctx.view_graph_filter_cycles(pairs, q)
    dt = (time.perf_counter() - t0) / reps
    out = {"metric": "view-graph edges filtered per second (rotation-cycle filter over all triplets)", "value": len(pairs) / dt, "unit": "edges/s",
           "images": a.images, "neighbors": a.neighbors, "edges": int(len(pairs)), "triplets": int(nt), "triplets_per_s": nt / dt,
           "seconds_per_call": dt, "kept_fraction": float(keep.mean()), "corrupted_edges": int(bad.sum()),
           "corrupted_edges_kept": int((keep & bad).sum()), "data": "synthetic", "note": "host buffers in and out: includes the host-side CSR build and both copies"}
    if a.cpu_edges > 0:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from tests.test_view_graph import _oracle_filter
        n_sub = max(4, min(a.images, int(a.cpu_edges / max(1, a.neighbors // 2)) + a.neighbors // 2))
        sel = (pairs[:, 0] < n_sub) & (pairs[:, 1] < n_sub)
        sp, sq = pairs[sel], q[sel]
        t0 = time.perf_counter()
        okeep, ont, _ = _oracle_filter(sp, sq)
        cdt = time.perf_counter() - t0
        dkeep, dnt = ctx.view_graph_filter_cycles(sp, sq)
        out["cpu_baseline"] = {"value": len(sp) / cdt, "unit": "edges/s", "cores": 1, "kind": "port", "triplets_per_s": ont / cdt,
                               "sample": "%d images, %d edges, %d triplets in %.2f s (oracle/view_graph.cc, 1 thread)" % (n_sub, len(sp), ont, cdt),
                               "device_equals_oracle_on_sample": bool((okeep == dkeep).all() and ont == dnt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
