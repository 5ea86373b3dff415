#!/bin/bash
out=gpurun_out/r3i
mkdir -p $out
export TMPDIR=/tmp
echo skip tests
cat > /tmp/knobs.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import tools.exp_verify_knobs as k
PY
timeout 900 python - > $out/tail_modes.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
from multiprocessing import Pool
import numpy as np
from dagsfm_amd import capi, sharding, synthetic
import tools.exp_verify_knobs as K
K._init()
with Pool(48, initializer=K._init) as pool:
    ims = pool.map(K._im, range(500), chunksize=4)
pairs = synthetic.exhaustive_pairs(500)
S = K._S
cams = [capi.simple_pinhole(S.focal, S.width / 2.0, S.height / 2.0, S.width, S.height, 1) for _ in range(500)]
ctx = capi.Context(0)
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
opts = capi.default_two_view_options()
combos = [dict(), dict(DSM_VERIFY_ITEM_MODE="1"), dict(DSM_LO_TAIL="4096"), dict(DSM_LO_TAIL="8192"), dict(DSM_LO_TAIL="16384"), dict(DSM_LO_TAIL="32768"), dict(DSM_LO_TAIL="8192", DSM_VERIFY_GRID_DIV="2"), dict(DSM_VERIFY_ITEM_MODE="1", DSM_VERIFY_GRID_DIV="2")]
keys = sorted({k for c in combos for k in c})
import ctypes
ref = None
for name, pl in (("1/8 shard", sharding.shard(pairs, 0, 8)), ("1/4 shard", sharding.shard(pairs, 0, 4)), ("1/2 shard", sharding.shard(pairs, 0, 2)), ("4950 pairs", pairs[:4950]), ("whole list", pairs)):
    ctx.match_pairs(pl)
    for c in combos + ([dict(DSM_VERIFY_INLINE_LO="0"), dict(DSM_VERIFY_INLINE_LO="1"), dict(DSM_VERIFY_INLINE_LO="0", DSM_VERIFY_ITEM_MODE="1")] if name == "1225 pairs" else []):
        for k in keys + ["DSM_VERIFY_INLINE_LO", "DSM_VERIFY_ITEM_MODE"]:
            os.environ.pop(k, None)
        os.environ.update(c)
        ctx.verify_pairs(opts, user_seed=0)
        best = 1e9
        for _ in range(2):
            ctx.verify_pairs(opts, user_seed=0)
            best = min(best, ctx.verify_kernel_time())
        recs = np.zeros((len(pl), ctypes.sizeof(capi.TwoViewGeometry)), dtype=np.uint8)
        capi.lib().dsm_get_two_view_geometries(ctx._h, recs.ctypes.data)
        if not c:
            ref = recs.copy()
        same = ref.shape == recs.shape and (ref == recs).all()
        print("%-10s %-50s verification %.1f ms   identical to default: %s" % (name, " ".join("%s=%s" % kv for kv in sorted(c.items())) or "default", best, same), flush=True)
PY
cat $out/tail_modes.txt
