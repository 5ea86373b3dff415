"""world_size-2 gloo test (CPU) of the multi-GPU path: pair sharding + two-phase all-gather of the
per-pair results reproduces the single-process result list."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_results(pairs):
    """Deterministic stand-in for per-pair device results: count = f(pair), rows = that many [i, j] entries."""
    counts = (pairs[:, 0] * 7 + pairs[:, 1] * 3) % 5
    rows = [np.stack([np.full(c, a), np.arange(c) + b], axis=1) for (a, b), c in zip(pairs, counts)]
    rows = np.concatenate(rows, axis=0) if len(rows) else np.zeros((0, 2), np.int64)
    return counts.astype(np.int32), rows.astype(np.int32).reshape(-1, 2)


def _worker(rank, world, port, q):
    from dagsfm_amd import sharding, synthetic
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pairs = synthetic.exhaustive_pairs(9).astype(np.int64)
    mine = sharding.shard(pairs, rank, world)
    counts, rows = _fake_results(mine)
    maxp = int(np.diff(sharding.shard_bounds(len(pairs), world)).max())
    allc = sharding.all_gather_fixed(dist, torch.from_numpy(counts).reshape(-1, 1), maxp, world)
    sizes, allr = sharding.all_gather_ragged(dist, torch.from_numpy(rows), world)
    b = sharding.shard_bounds(len(pairs), world)
    got_counts = torch.cat([allc[r * maxp:r * maxp + int(b[r + 1] - b[r])] for r in range(world)]).numpy().reshape(-1)
    got_rows = sharding.assemble_ragged(sizes, allr).numpy()
    q.put((rank, got_counts, got_rows))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_two_ranks():
    from dagsfm_amd import sharding, synthetic
    pairs = synthetic.exhaustive_pairs(9).astype(np.int64)
    ref_counts, ref_rows = _fake_results(pairs)
    # shards are a partition in list order
    parts = [sharding.shard(pairs, r, 2) for r in range(2)]
    assert (np.concatenate(parts) == pairs).all()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, c, rows in res:
        assert (c == ref_counts).all()
        assert rows.shape == ref_rows.shape and (rows == ref_rows).all()


def test_shard_bounds_cover_uneven():
    from dagsfm_amd import sharding
    for n in (0, 1, 7, 8, 9, 124750):
        for w in (1, 2, 4, 8):
            b = sharding.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1
