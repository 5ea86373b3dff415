"""The first RCCL call happens here, not in the driver's SCALE run.

A one-GPU box cannot host two RCCL ranks, but ONE rank can form a process group on backend "nccl" (= RCCL on ROCm) bound
to the device, and `sharding.gather_match_graph(..., force_collectives=...)` then sends the real results of a real
context -- int64 counts, uint8 TwoViewGeometry records, int32 match rows, all in device memory, fetched through the
C-ABI's device-pointer getters -- through `all_gather_into_tensor` and through the per-rank `broadcast` of the skewed
exchange.  The graph must equal the one assembled without any collective, byte for byte.  bench.py --force-collectives
is the same path with the exchange timed (the reference analogue: one matcher per device, results merged by the caller,
/root/reference/src/feature/matching.cc:631-645)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from dagsfm_amd import capi, sharding, synthetic
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
n_img = 14
scene = synthetic.Scene(n_img, 640, seed=3)
ims = [scene.image(i) for i in range(n_img)]
pairs = synthetic.exhaustive_pairs(n_img)
cams = [capi.simple_pinhole(scene.focal, scene.width / 2.0, scene.height / 2.0, scene.width, scene.height, True) for _ in range(n_img)]
ctx = capi.Context(0)
ctx.set_images([im[0] for im in ims], [im[1] for im in ims], cams)
ctx.match_pairs(pairs)
ctx.verify_pairs(capi.default_two_view_options(), user_seed=0, stage_filter=True)
src = sharding.CtxSource(ctx, len(pairs), dev)
bounds = sharding.shard_bounds(len(pairs), 1)
graphs = {}
for name, force in (("plain", None), ("auto", True), ("padded", "padded"), ("broadcast", "broadcast")):
    g = sharding.gather_match_graph(dist, src, 0, 1, bounds, True, force_collectives=force)
    torch.cuda.synchronize()
    for t in (g.match_counts, g.matches, g.tvg, g.inlier_counts, g.inlier_matches):
        assert t.is_cuda
    graphs[name] = [t.cpu().numpy() for t in (g.match_counts, g.matches, g.tvg, g.inlier_counts, g.inlier_matches)]
# a plain all_reduce on the same group, the call bench.py uses for the max-over-ranks time
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
ok = all(a.shape == b.shape and (a == b).all() for k in ("auto", "padded", "broadcast") for a, b in zip(graphs[k], graphs["plain"]))
print(json.dumps({"ok": bool(ok), "pairs": int(len(pairs)), "matches": int(graphs["plain"][1].shape[0]),
                  "inlier_matches": int(graphs["plain"][4].shape[0]), "all_reduce": float(t.item()), "backend": dist.get_backend()}))
dist.destroy_process_group()
"""


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_rank_rccl_gathers_the_same_graph(tmp_path):
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_WORKER % {"root": ROOT, "port": _free_port()})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["backend"] == "nccl" and out["all_reduce"] == 3.5
    assert out["pairs"] == 91 and out["matches"] > 500 and out["inlier_matches"] > 200


def test_bench_force_collectives_reports_the_exchange(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--images", "40", "--feats", "1024", "--steps", "2", "--warmup", "1",
           "--cpu-seconds", "0", "--force-collectives"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ex = d["exchange"]
    assert ex["backend"] == "nccl" and ex["forced_on_one_rank"] and d["n_gpus"] == 1
    assert ex["gather_ms_per_step"] > 0 and ex["bytes_per_step"] > 0
