"""The multi-rank path with REAL dsm contexts: bench.py --gpus 2 --oversubscribe runs two ranks
on device 0, each with its own context over its cost-cut share of the pair list, and assembles the match graph through
sharding.gather_match_graph -- CtxSource's device-pointer getters included -- over gloo (RCCL refuses two ranks on one
GPU; the collective calls are the same).  The assembled graph must equal the single-rank graph byte for byte.
Unmeasured on 8 GPUs: only the driver's SCALE run has the node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(tmp_path, tag, extra):
    out = str(tmp_path / (tag + ".npz"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--images", "28", "--feats", "640", "--steps", "1", "--warmup", "0",
           "--cpu-seconds", "0", "--dump-graph", out] + extra
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), dict(np.load(out))


@pytest.mark.parametrize("pairs,cut", [("exhaustive", "interleaved"), ("knn:6", "interleaved"), ("exhaustive", "contiguous")])
def test_two_ranks_assemble_the_single_rank_graph(tmp_path, pairs, cut):
    one, g1 = _bench(tmp_path, "one", ["--pairs", pairs])
    two, g2 = _bench(tmp_path, "two", ["--pairs", pairs, "--gpus", "2", "--oversubscribe", "--cut", cut])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and cut in two["config"]["parallelism"]
    # what every rank did (min, mean, max over the ranks): both ranks had pairs, and the slowest rank's step is the job's
    pr = two["per_rank"]
    assert pr["pairs"][0] > 0 and sum(pr["pairs"]) > 0 and pr["ms_per_step"][2] <= two["ms_per_step"] * 1.5
    assert "per_rank" not in one
    assert one["config"]["pairs"] == two["config"]["pairs"] == len(g1["match_counts"])
    assert one["config"]["pairs_with_geometry"] > 10
    for k in ("match_counts", "matches", "tvg", "inlier_counts", "inlier_matches"):
        assert g1[k].shape == g2[k].shape and (g1[k] == g2[k]).all(), k
