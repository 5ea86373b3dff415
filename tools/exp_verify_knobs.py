#!/usr/bin/env python3
"""Verification time of config 2 and of its 1/8 shard under the scheduling knobs (environment variables, read per call):
lanes, persistent-grid divisor, inline-tail length.  One process, one scene, the matches computed once per list."""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, sharding, synthetic  # noqa: E402

_S = None


def _init():
    global _S
    _S = synthetic.Scene(500, 4096, seed=0)


def _im(i):
    return _S.image(i)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--combos", default="", help='semicolon-separated "KEY=V KEY=V" settings instead of the built-in list')
    a = ap.parse_args()
    _init()
    with Pool(48, initializer=_init) as pool:
        ims = pool.map(_im, range(500), chunksize=4)
    pairs = synthetic.exhaustive_pairs(500)
    cams = [capi.simple_pinhole(_S.focal, _S.width / 2.0, _S.height / 2.0, _S.width, _S.height, 1) for _ in range(500)]
    ctx = capi.Context(0)
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    opts = capi.default_two_view_options()
    combos = [dict(), dict(DSM_VERIFY_LANES="1"), dict(DSM_VERIFY_LANES="3"), dict(DSM_VERIFY_LANES="4"),
              dict(DSM_VERIFY_LANES="2", DSM_VERIFY_GRID_DIV="2"), dict(DSM_VERIFY_LANES="3", DSM_VERIFY_GRID_DIV="3"),
              dict(DSM_VERIFY_LANES="4", DSM_VERIFY_GRID_DIV="4"), dict(DSM_VERIFY_LANES="4", DSM_VERIFY_GRID_DIV="2"),
              dict(DSM_LO_TAIL="512"), dict(DSM_LO_TAIL="4096"), dict(DSM_LO_TAIL="16384"), dict(DSM_LO_TAIL="32768"),
              dict(DSM_VERIFY_LANES="2", DSM_VERIFY_LANE_SPLIT="0.6"), dict(DSM_VERIFY_LANES="2", DSM_VERIFY_LANE_SPLIT="0.7"),
              dict(DSM_VERIFY_ITEM_MODE="1"), dict(DSM_VERIFY_ITEM_MODE="0"), dict(DSM_VERIFY_LANES="3", DSM_VERIFY_GRID_DIV="2"),
              dict(DSM_VERIFY_LANES="4", DSM_VERIFY_GRID_DIV="4", DSM_LO_TAIL="512")]
    if a.combos:
        combos = [dict()] + [dict(kv.split("=") for kv in c.split()) for c in a.combos.split(";") if c.strip()]
    keys = sorted({k for c in combos for k in c})
    for name, pl in (("1/8 shard", sharding.shard(pairs, 0, 8)), ("whole list", pairs)):
        ctx.match_pairs(pl)
        for c in combos:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(c)
            ctx.verify_pairs(opts, user_seed=0)
            best = 1e9
            for _ in range(2):
                t = time.perf_counter()
                ctx.verify_pairs(opts, user_seed=0)
                ctx.sync()
                wall = (time.perf_counter() - t) * 1e3
                best = min(best, ctx.verify_kernel_time())
            print("%-10s %-70s verification %.1f ms (wall of the call %.1f)" % (name, " ".join("%s=%s" % kv for kv in sorted(c.items())) or "default", best, wall), flush=True)
        for k in keys:
            os.environ.pop(k, None)


if __name__ == "__main__":
    main()
