#!/usr/bin/env python3
"""What the interleaved cut's reorder costs (sharding.gather_match_graph, order=...): the gathered results of config 2 arrive rank by
rank and are put back into list order on the device.  One GPU: the rank-major arrays are built from a single-rank graph, then the
reorder is timed (the same torch calls the N-rank exchange makes after its collectives)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, sharding, synthetic  # noqa: E402

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scene = synthetic.Scene(n_img, 4096, seed=0)
ims = [scene.image(i) for i in range(n_img)]
cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, True) for _ in range(n_img)]
pairs = synthetic.exhaustive_pairs(n_img)
ctx = capi.Context(0)
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
ctx.match_pairs(pairs)
ctx.verify_pairs(capi.default_two_view_options(), user_seed=0, stage_filter=True)
dev = torch.device("cuda", 0)
g = sharding.gather_match_graph(None, sharding.CtxSource(ctx, len(pairs), dev), 0, 1, sharding.shard_bounds(len(pairs), 1), True)
parts = sharding.interleaved_parts(len(pairs), 8)
bounds, order = sharding.parts_bounds_and_order(parts)
o = torch.as_tensor(order, device=dev)
# rank-major arrays = what the collectives deliver
mc_rm, m_rm = sharding.reorder_rows(g.match_counts, g.matches, o)
ic_rm, i_rm = sharding.reorder_rows(g.inlier_counts, g.inlier_matches, o)
tvg_rm = g.tvg[o]
pos = torch.empty(len(order), dtype=torch.int64)
pos[torch.as_tensor(order)] = torch.arange(len(order), dtype=torch.int64)
for rep in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    p = pos.to(dev)
    mc, m = sharding.reorder_rows(mc_rm, m_rm, p)
    tv = tvg_rm[p]
    ic, im = sharding.reorder_rows(ic_rm, i_rm, p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ok = bool((mc == g.match_counts).all() and (m == g.matches).all() and (tv == g.tvg).all() and (im == g.inlier_matches).all())
    print("reorder of %d pairs, %d + %d rows: %.2f ms, back in list order: %s" % (len(pairs), len(m), len(im), 1e3 * dt, ok))
