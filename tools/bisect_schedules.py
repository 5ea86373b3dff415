#!/usr/bin/env python3
"""Which ingredient of the batched schedule makes it differ from the inline one (debug)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagsfm_amd import capi, synthetic

def run(ctx, opts, env):
    for k in ("DSM_VERIFY_LEGACY", "DSM_VERIFY_LANES", "DSM_LO_TAIL", "DSM_LO_JACOBI_GROUPS", "DSM_LO_PREPARE_WAVE", "DSM_VERIFY_FIXED_BATCH"):
        os.environ.pop(k, None)
    os.environ["DSM_VERIFY_INLINE_LO"] = "0"
    os.environ.update(env)
    ctx.verify_pairs(opts, user_seed=0, stage_filter=True)
    recs = np.zeros((ctx.n_pairs, ctypes.sizeof(capi.TwoViewGeometry)), dtype=np.uint8)
    assert ctx._L.dsm_get_two_view_geometries(ctx._h, recs.ctypes.data) == 0
    return recs, [t for t in ctx.two_view_geometries()]

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scene = synthetic.Scene(n_img, 4096, seed=0)
ims = [scene.image(i) for i in range(n_img)]
pairs = synthetic.exhaustive_pairs(n_img)
ctx = capi.Context(0, check=True)
cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, 1) for _ in range(n_img)]
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
ctx.match_pairs(pairs)
opts = capi.default_two_view_options()
ref, ref_t = run(ctx, opts, {"DSM_VERIFY_INLINE_LO": "1"})
for name, env in [("default", {}), ("jacobi groups", {"DSM_LO_JACOBI_GROUPS": "1"}), ("wave prepare", {"DSM_LO_PREPARE_WAVE": "1"}),
                  ("no tail", {"DSM_LO_TAIL": "0"}), ("fixed batch", {"DSM_VERIFY_FIXED_BATCH": "1"}),
                  ("all old", {"DSM_LO_JACOBI_GROUPS": "1", "DSM_LO_PREPARE_WAVE": "1", "DSM_LO_TAIL": "0", "DSM_VERIFY_FIXED_BATCH": "1"})]:
    got, got_t = run(ctx, opts, env)
    bad = np.nonzero((ref != got).any(axis=1))[0]
    print("%-14s differing pairs: %d %s" % (name, len(bad), bad[:6]), flush=True)
    for k in bad[:2]:
        a, b = ref_t[k], got_t[k]
        print("   pair", k, "config", a.config, b.config, "inliers", a.num_inliers, b.num_inliers, "trials", list(a.num_trials), list(b.num_trials),
              "models", list(a.num_models), list(b.num_models))
        for nm in ("E", "F", "H"):
            x, y = np.array(list(getattr(a, nm))), np.array(list(getattr(b, nm)))
            if (x != y).any():
                print("     ", nm, "max abs diff %.3e" % np.abs(x - y).max())
