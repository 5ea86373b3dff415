"""libdagsfm_gather.so (include/dagsfm_gather.h): the RCCL assembly of the match graph BELOW the host language (missing 4).

A one-GPU box can hold ONE rank: dsm_gather_create then builds a one-device communicator with ncclCommInitAll and
dsm_gather_match_graph sends a real context's results through the same grouped ncclAllGather (offsets, records) and in-place
ncclBroadcast (matches, inlier matches) calls an n-device host makes, followed by the same compaction kernel.  The assembled graph
must equal what the context's own getters return, byte for byte; a second call with other sizes re-uses the buffers; a list
without geometry and an empty list work; two contexts on ONE device are refused (RCCL does not take a device twice).  Runs in a
subprocess so that librccl is mapped next to nothing else of the suite.  Reference analogue: one matcher per gpu_index device,
outputs merged by the caller (/root/reference/src/feature/matching.cc:631-645, 814-836)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_WORKER = r"""
import ctypes, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from dagsfm_amd import capi, synthetic
res = {}
n_img = 14
scene = synthetic.Scene(n_img, 640, seed=3)
ims = [scene.image(i) for i in range(n_img)]
cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, True) for _ in range(n_img)]
ctx = capi.Context(0, check=False)  # (the companion library links the product build)
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
g = capi.Gather([ctx])
ok = True
for n_use in (n_img, 5):
    pairs = synthetic.exhaustive_pairs(n_use)
    ctx.match_pairs(pairs)
    ctx.verify_pairs(capi.default_two_view_options(), user_seed=0, stage_filter=True)
    moff, m, tv, ioff, im = g.match_graph([len(pairs)], True)
    o0, m0 = ctx.matches()
    t0 = ctx.two_view_geometries()
    i0, im0 = ctx.inlier_matches()
    ok = ok and (moff == o0).all() and (m == m0).all() and (ioff == i0).all() and (im == im0).all() and len(tv) == len(t0) == len(pairs)
    ok = ok and all(bytes(a) == bytes(b) for a, b in zip(tv, t0))
    res["pairs_%%d" %% n_use] = [int(len(pairs)), int(len(m)), int(len(im)), round(g.time_ms(), 3)]
    # without geometry: offsets and matches only
    moff2, m2 = g.match_graph([len(pairs)], False)
    ok = ok and (moff2 == o0).all() and (m2 == m0).all()
# the assembled arrays as device pointers (what an on-device consumer reads)
ptrs = [ctypes.c_void_p() for _ in range(5)]
rc = capi.gather_lib().dsm_gather_device_arrays(g._g, 0, *[ctypes.byref(p) for p in ptrs])
ok = ok and rc == 0 and ptrs[0].value and ptrs[1].value and not ptrs[2].value  # (the last call was without geometry)
# a share length that is not the context's is refused (it would overrun the staging blocks)
try:
    g.match_graph([len(pairs) + 3], False)
    ok = False
except capi.DsmError:
    pass
# an empty share list
moffe, me = g.match_graph([0], False)
ok = ok and len(moffe) == 1 and moffe[0] == 0 and len(me) == 0
g.close()
# one device twice: refused up front
other = capi.Context(0, check=False)
try:
    capi.Gather([ctx, other])
    ok = False
except capi.DsmError:
    pass
res["ok"] = bool(ok)
print(json.dumps(res))
"""


def test_gather_library_assembles_the_graph_through_rccl():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])  # (RCCL prints its own "Librccl path" line to stdout)
    assert out["ok"], out
    assert out["pairs_14"][0] == 91 and out["pairs_14"][1] > 1000 and out["pairs_14"][2] > 500
    # librccl was mapped by the companion library, not by the product library
    maps = subprocess.run(["ldd", os.path.join(ROOT, "dagsfm_amd", "libdagsfm_mi355x.so")], capture_output=True, text=True).stdout
    assert "rccl" not in maps
