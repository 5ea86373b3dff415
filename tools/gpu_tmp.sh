mkdir -p gpurun_out/r3j
timeout 900 python - > gpurun_out/r3j/debug.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from multiprocessing import Pool
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
pl = sharding.shard(pairs, 0, 8)
ctx.match_pairs(pl)
os.environ["DSM_VERIFY_LANES"] = "1"
for c in (dict(DSM_LO_TAIL="0"), dict(DSM_LO_TAIL="512"), dict(DSM_LO_TAIL="2048"), dict(DSM_LO_TAIL="2048", DSM_LO_TAIL_MODE="inline")):
    for k in ("DSM_LO_TAIL", "DSM_LO_TAIL_MODE"):
        os.environ.pop(k, None)
    os.environ.update(c)
    ctx.verify_pairs(opts, user_seed=0)
    os.environ["DSM_VERIFY_DEBUG"] = "1"
    sys.stderr.write("=== %s\n" % c); sys.stderr.flush()
    ctx.verify_pairs(opts, user_seed=0)
    os.environ.pop("DSM_VERIFY_DEBUG")
    sys.stderr.write("verification %.1f ms\n" % ctx.verify_kernel_time()); sys.stderr.flush()
PY
cat gpurun_out/r3j/debug.txt
