"""Small profiling driver: 60 images x 4096 features, exhaustive pairs, calibrated verification (3 runs)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scene = synthetic.Scene(n_img, 4096, seed=0)
ims = [scene.image(i) for i in range(n_img)]
pairs = synthetic.exhaustive_pairs(n_img)
ctx = capi.Context(0)
cams = [capi.simple_pinhole(800., 500., 375., 1000, 750, 1) for _ in range(n_img)]
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
ctx.match_pairs(pairs)
o = capi.default_two_view_options()
for _ in range(3):
    ctx.verify_pairs(o)
    print('verify ms %.2f for %d pairs' % (ctx.verify_kernel_time(), len(pairs)), flush=True)
