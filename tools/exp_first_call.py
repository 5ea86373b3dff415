#!/usr/bin/env python3
"""Where does the FIRST Match() of a process spend its time (the drop-in CLI's one-shot run: device 1.6 s against 0.56 s warm)?
Times set_images / match_pairs / verify_pairs of call 1, 2, 3 on config 2 in one process.
    python tools/exp_first_call.py [--images 500]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dagsfm_amd import capi, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=500)
a = ap.parse_args()
scene = synthetic.Scene(a.images, 4096, seed=0)
ims = [scene.image(i) for i in range(a.images)]
pairs = synthetic.exhaustive_pairs(a.images)
cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, 1) for _ in range(a.images)]
t0 = time.perf_counter()
ctx = capi.Context(0)
t1 = time.perf_counter()
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
ctx.sync()
t2 = time.perf_counter()
print("context %.3f s, set_images %.3f s" % (t1 - t0, t2 - t1), flush=True)
opts, topts = capi.default_match_options(), capi.default_two_view_options()
for k in range(3):
    ta = time.perf_counter()
    ctx.match_pairs(pairs, opts)
    ctx.sync()
    tb = time.perf_counter()
    ctx.verify_pairs(topts, user_seed=0, stage_filter=True)
    ctx.sync()
    tc = time.perf_counter()
    offs, m = ctx.matches()
    tv = ctx.two_view_geometries()
    io, im = ctx.inlier_matches()
    td = time.perf_counter()
    print("call %d: match %.3f s (kernels %.3f), verify %.3f s (kernels %.3f), fetch to host %.3f s" % (
        k + 1, tb - ta, ctx.match_kernel_time()[0] / 1e3 + ctx.match_gather_time() / 1e3 + ctx.match_resolve_time() / 1e3, tc - tb,
        ctx.verify_kernel_time() / 1e3, td - tc), flush=True)
