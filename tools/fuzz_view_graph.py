#!/usr/bin/env python3
"""Differential fuzz of the view-graph cycle filter (dsm_view_graph_filter_cycles) against oracle/view_graph.cc: seeded
random graphs of awkward shapes -- no edge, one edge, cliques, stars (no triangle), duplicated and swapped edges, huge
sparse ids, corrupted rotations, noise -- keep flags and triplet counts must agree (a case with a loop error within
1e-6 degrees of the threshold is skipped: the device's angle is not bit-identical there).

  python tools/fuzz_view_graph.py [--cases 200] [--seed 1]

Test infrastructure: the oracle is the checker here, as in tests/."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi  # noqa: E402


def run_fuzz(ctx, n_cases, seed, log=print):
    from tests.test_view_graph import _graph, _oracle_filter, _qconj, _rand_rot
    bad = total = skipped = 0
    for c in range(n_cases):
        rng = np.random.default_rng([seed, c])
        kind = str(rng.choice(["random", "random", "clique", "star", "empty", "single", "path"]))
        n_img = int(rng.choice([2, 3, 4, 6, 12, 50, 300]))
        if kind == "clique":
            n_img = min(n_img, 12)
            pairs = [(i, j) for i in range(n_img) for j in range(i + 1, n_img)]
        elif kind == "star":
            pairs = [(0, j) for j in range(1, n_img)]
        elif kind == "path":
            pairs = [(j, j + 1) for j in range(n_img - 1)]
        elif kind == "empty":
            pairs = []
        elif kind == "single":
            pairs = [(0, 1)]
        else:
            deg = int(rng.choice([1, 2, 4, 10]))
            s = set()
            for i in range(n_img):
                for j in rng.choice(n_img, min(deg, n_img - 1), replace=False):
                    if i != int(j):
                        s.add((min(i, int(j)), max(i, int(j))))
            pairs = sorted(s)
        pairs = [pairs[k] for k in rng.permutation(len(pairs))]
        n_cor = int(rng.choice([0, 0, 1, 5, 50]))
        noise = float(rng.choice([0.0, 0.0, 0.01, 0.05]))
        thr = float(rng.choice([1.0, 5.0, 20.0]))
        q = _graph(rng, n_img, pairs, corrupt=rng.choice(len(pairs), min(n_cor, len(pairs)), replace=False) if pairs else (), noise=noise) \
            if pairs else np.zeros((0, 4))
        ids = rng.permutation(int(rng.choice([n_img, 10 * n_img, 2**31 - 2])) if n_img > 300 else max(n_img, int(rng.choice([n_img, 10 * n_img, 100000]))))[:n_img] + 1
        pid = [(int(ids[a]), int(ids[b])) for a, b in pairs]
        qs = np.array([qq if x < y else _qconj(qq) for (x, y), qq in zip(pid, q)]).reshape(-1, 4)
        pid = [(min(x, y), max(x, y)) for x, y in pid]
        if len(pid) > 2 and rng.random() < 0.3:  # duplicates: the first occurrence counts
            k = int(rng.integers(len(pid)))
            pid.append(pid[k])
            qs = np.vstack([qs, _rand_rot(rng, 1)])
        ref_keep, ref_n, ref_err = _oracle_filter(pid, qs, thr)
        fin = ref_err[np.isfinite(ref_err)]
        if len(fin) and np.abs(fin - thr).min() <= 1e-6:
            skipped += 1
            continue
        keep, n = ctx.view_graph_filter_cycles(np.array(pid, np.uint32).reshape(-1, 2), qs, thr)
        total += 1
        if n != ref_n or not (keep == ref_keep).all():
            bad += 1
            log("MISMATCH case %d (%s, %d images, %d edges, thr %.0f): triplets %d vs %d, keep differs at %s" %
                (c, kind, n_img, len(pid), thr, n, ref_n, np.nonzero(keep != ref_keep)[0][:8].tolist()))
    return total, bad, skipped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    ctx = capi.Context(0)
    total, bad, skipped = run_fuzz(ctx, args.cases, args.seed, log=lambda s: print(s, flush=True))
    print("FUZZ RESULT: %d graphs (%d skipped near the threshold), %d mismatches" % (total, skipped, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
