#!/usr/bin/env python3
"""Differential fuzz of the C++ host shim (dagsfm_amd/host: ExhaustiveFeatureMatcher::Run -> SiftFeatureMatcher::Match ->
database.db) against the oracle: seeded databases with very uneven images (0, a handful, a few hundred features),
random block sizes (so that Match() is called many times with pair lists of changing length on one context, and pairs
are visited as (larger id, smaller id)), calibrated or not, write-back on the caller or on a background thread -- every
`matches` and `two_view_geometries` row against the oracle, then a second run that must change nothing (resume).

  python tools/fuzz_host.py [--cases 20] [--seed 1]

Test infrastructure: the oracle is the checker here, as in tests/."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import capi, synthetic  # noqa: E402
from tests import dbutil  # noqa: E402

CLI = os.path.join(ROOT, "dagsfm_amd", "dsm_exhaustive_matcher")


def visit_order(n, block_size):
    """ExhaustiveFeatureMatcher::Run's block loop, /root/reference/src/feature/matching.cc:870-905 (0-based indices)."""
    out = []
    for s1 in range(0, n, block_size):
        e1 = min(n, s1 + block_size) - 1
        for s2 in range(0, n, block_size):
            e2 = min(n, s2 + block_size) - 1
            for i1 in range(s1, e1 + 1):
                for i2 in range(s2, e2 + 1):
                    b1, b2 = i1 % block_size, i2 % block_size
                    if (i1 > i2 and b1 <= b2) or (i1 < i2 and b1 < b2):
                        out.append((i1, i2))
    return out


def quat_inverse_pose(rq, rt):
    w, x, y, z = rq[0], -rq[1], -rq[2], -rq[3]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return np.array([w, x, y, z]), -R @ rt


def run_fuzz(n_cases, seed, log=print, tmpdir=None):
    from tests import oracle_lib
    oracle = oracle_lib.load()
    bad = total = 0
    for c in range(n_cases):
        rng = np.random.default_rng([seed, c])
        n_img = int(rng.integers(3, 13))
        feats = int(rng.choice([120, 300, 500]))
        prior = bool(rng.random() < 0.6)
        block = int(rng.integers(2, n_img + 3))
        async_write = bool(rng.random() < 0.5)
        user_seed = int(rng.integers(0, 1000))
        scene = synthetic.Scene(n_img, feats, seed=int(rng.integers(0, 2**31)), n_pool=int(feats * rng.uniform(1.1, 2.5)),
                                planar=bool(rng.random() < 0.2))
        ims = []
        for i in range(n_img):
            d, k = scene.image(i)[:2]
            r = rng.random()
            n = len(d) if r < 0.55 else (0 if r < 0.65 else int(rng.choice([3, 10, 20, 60])))
            ims.append((np.ascontiguousarray(d[:n]), np.ascontiguousarray(k[:n])))
        with tempfile.TemporaryDirectory(dir=tmpdir) as td:
            path = os.path.join(td, "database.db")
            dbutil.create(path, ims, prior=prior)
            env = dict(os.environ)
            env.pop("DSM_ASYNC_WRITE_BACK", None)
            if async_write:
                env["DSM_ASYNC_WRITE_BACK"] = "1"
            cmd = [CLI, "--database_path", path, "--ExhaustiveMatching.block_size", str(block), "--random_seed", str(user_seed)]
            subprocess.run(cmd, check=True, env=env, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            matches, tvgs = dbutil.read_results(path)
            subprocess.run(cmd, check=True, env=env, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            matches2, tvgs2 = dbutil.read_results(path)
        cam = capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, prior)
        opts = capi.default_two_view_options()
        order = visit_order(n_img, block)
        nb = 0
        if len(order) != n_img * (n_img - 1) // 2 or len(matches) != len(order) or len(tvgs) != len(order):
            nb += 1
            log("MISMATCH case %d: %d pairs visited, %d matches rows, %d geometry rows" % (c, len(order), len(matches), len(tvgs)))
        for a, b in order:
            pid = dbutil.pair_id(a + 1, b + 1)
            d = None
            ref_m = oracle.match_sift_features_cpu(ims[a][0], ims[b][0])
            if len(ref_m) < 15:
                ref_m = np.zeros((0, 2), np.uint32)  # matching.cc:430-432: fewer matches than min_num_inliers are dropped
            swap = a > b
            if pid not in matches or pid not in tvgs:
                d = "row missing"
            elif not (matches[pid].shape == ref_m.shape and (matches[pid] == (ref_m[:, ::-1] if swap else ref_m)).all()):
                d = "matches"
            else:
                ref, ref_inl = oracle.estimate_two_view_geometry(cam, ims[a][1].astype(np.float64), cam, ims[b][1].astype(np.float64),
                                                                 ref_m, opts, capi.pair_seed(a + 1, b + 1, user_seed))
                t = tvgs[pid]
                if ref.num_inliers >= 15:
                    rq, rt = np.array(list(ref.qvec)), np.array(list(ref.tvec))
                    if swap:
                        rq, rt = quat_inverse_pose(rq, rt)
                    exp_inl = ref_inl[:, ::-1] if swap else ref_inl
                    if t["config"] != ref.config:
                        d = "config %d vs %d" % (t["config"], ref.config)
                    elif not (t["inliers"].shape == exp_inl.shape and (t["inliers"] == exp_inl).all()):
                        d = "inlier matches"
                    elif not (np.allclose(np.frombuffer(t["F"], np.float64), rq, rtol=1e-6, atol=1e-12) and
                              np.allclose(np.frombuffer(t["E"], np.float64), rt, rtol=1e-6, atol=1e-9)):
                        d = "pose"
                elif not (t["config"] == 0 and len(t["inliers"]) == 0):
                    d = "filtered pair has config %d, %d inliers" % (t["config"], len(t["inliers"]))
                if d is None:
                    t2 = tvgs2.get(pid)
                    if t2 is None or t2["config"] != t["config"] or t2["F"] != t["F"] or not (t2["inliers"] == t["inliers"]).all() or \
                            not (matches2[pid] == matches[pid]).all():
                        d = "second run changed the rows"
            if d is not None:
                nb += 1
                if nb <= 5:
                    log("MISMATCH case %d pair (%d,%d): %s" % (c, a + 1, b + 1, d))
        bad += nb
        total += len(order)
        log("case %d: %d images %s, block_size %d, prior %d, async %d: %d pairs, %d mismatches" %
            (c, n_img, [len(im[0]) for im in ims], block, prior, async_write, len(order), nb))
    return total, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    total, bad = run_fuzz(args.cases, args.seed, log=lambda s: print(s, flush=True))
    print("FUZZ RESULT: %d pairs, %d mismatches" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
