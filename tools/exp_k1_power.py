#!/usr/bin/env python3
"""Is k1_best_rows limited by the chip's power budget rather than by its instruction stream?  The same binary, the
same launch (200 images x 4 096 rows, 19 900 pairs), descriptors of different bit activity: the MFMA count is identical,
only the data toggling differs (MI355X_MICROARCH.md, "DVFS give-back").  Prints pass-1 time per data set."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic  # noqa: E402

n_img, n_feat = 200, 4096
rng = np.random.default_rng(0)
scene = synthetic.Scene(n_img, n_feat, seed=0)
sets = {
    "synthetic SIFT (bench data)": lambda i: scene.image(i)[0],
    "uniform random u8": lambda i: rng.integers(0, 256, (n_feat, 128), dtype=np.uint8),
    "all 128 (s8 operand 0)": lambda i: np.full((n_feat, 128), 128, dtype=np.uint8),
    "all 0 (s8 operand -128)": lambda i: np.zeros((n_feat, 128), dtype=np.uint8),
    # the same SIFT-like bytes (all below 128 in this generator) seen by the multipliers as plain positive int8 operands instead of
    # as value - 128: does the ENCODING of small magnitudes matter, or only how much the low bits toggle?
    "synthetic SIFT + 128 (operand = the byte itself)": lambda i: (scene.image(i)[0].astype(np.int32) + 128).clip(0, 255).astype(np.uint8),
    "uniform 0..127 (operands -128..-1)": lambda i: rng.integers(0, 128, (n_feat, 128), dtype=np.uint8),
    "uniform 128..255 (operands 0..127)": lambda i: rng.integers(128, 256, (n_feat, 128), dtype=np.uint8),
    "uniform 0..15 (operands -128..-113)": lambda i: rng.integers(0, 16, (n_feat, 128), dtype=np.uint8),
    "uniform 128..143 (operands 0..15)": lambda i: rng.integers(128, 144, (n_feat, 128), dtype=np.uint8),
}
kp = np.zeros((n_feat, 2), dtype=np.float32)
cams = [capi.simple_pinhole(800.0, 500.0, 375.0, 1000, 750, 1) for _ in range(n_img)]
pairs = synthetic.exhaustive_pairs(n_img)
for name, gen in sets.items():
    ctx = capi.Context(0)
    ctx.set_images([gen(i) for i in range(n_img)], [kp] * n_img, cams)
    ctx.match_pairs(pairs)
    ctx.match_pairs(pairs)
    ms, n = ctx.match_kernel_time()   # pass-1 launches of the last call
    ms /= max(1, n)
    ops = 2.0 * 128 * n_feat * n_feat * len(pairs)
    print("%-50s pass 1 %.2f ms  %.2f POP/s  (%.3f of 5.03 POP/s)" % (name, ms, ops / ms / 1e12, ops / ms / 1e12 / 5.033), flush=True)
    del ctx
