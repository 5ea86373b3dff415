import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic
scene = synthetic.Scene(60, 4096, seed=0)
ims = [scene.image(i) for i in range(60)]
pairs = synthetic.exhaustive_pairs(60)
ctx = capi.Context(0)
for prior in (1, 0):
    cams = [capi.simple_pinhole(800., 500., 375., 1000, 750, prior) for _ in range(60)]
    ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
    ctx.match_pairs(pairs)
    for mt in (10000, 256, 64):
        o = capi.default_two_view_options(max_num_trials=mt)
        ctx.verify_pairs(o); ctx.verify_pairs(o)
        tv = ctx.two_view_geometries()
        tr = np.array([list(t.num_trials) for t in tv]).mean(0)
        mo = np.array([list(t.num_models) for t in tv]).mean(0)
        print('prior', prior, 'max_trials', mt, 'verify ms %.1f' % ctx.verify_kernel_time(), 'pairs', len(pairs), 'mean trials', tr.round(1), 'mean models', mo.round(1), flush=True)
