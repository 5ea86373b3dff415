"""world_size-2 gloo test (CPU) of the multi-GPU path: pair sharding + two-phase all-gather of the
per-pair results reproduces the single-process result list."""
import os
import socket
import sys

import numpy as np
import pytest
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


def _skew_worker(rank, world, port, q):
    from dagsfm_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    out = []
    for sizes in ((5, 400), (0, 7), (300, 290), (0, 0)):   # skewed -> per-rank broadcasts; balanced -> one padded all-gather
        n = sizes[rank]
        local = torch.arange(n * 2, dtype=torch.int32).reshape(n, 2) + 1000 * rank
        out.append(sharding.all_gather_rows(dist, local, rank, world).numpy().copy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_rows_exact_sizes_when_skewed():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_skew_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs in res:
        for sizes, got in zip(((5, 400), (0, 7), (300, 290), (0, 0)), outs):
            ref = np.concatenate([np.arange(n * 2, dtype=np.int32).reshape(n, 2) + 1000 * r for r, n in enumerate(sizes)])
            assert got.shape == ref.shape and (got == ref).all()


def test_cost_aware_shard_bounds():
    from dagsfm_amd import sharding
    rng = np.random.default_rng(0)
    nfeat = rng.integers(500, 9000, 60)
    pairs = np.array([(i, j) for i in range(60) for j in range(i + 1, 60)][::3])
    costs = sharding.pair_costs(pairs, nfeat)
    for w in (1, 2, 4, 8):
        b = sharding.shard_bounds(len(pairs), w, costs)
        assert b[0] == 0 and b[-1] == len(pairs) and (np.diff(b) >= 0).all()
        per = np.array([costs[b[r]:b[r + 1]].sum() for r in range(w)])
        assert per.max() <= costs.sum() / w + costs.max()          # no rank is more than one pair above its share
        assert (np.concatenate([sharding.shard(pairs, r, w, costs) for r in range(w)]) == pairs).all()
    # equal costs fall back to (almost) equal counts
    b = sharding.shard_bounds(1000, 8, np.full(1000, 3.0))
    assert np.diff(b).max() - np.diff(b).min() <= 1


def test_shard_bounds_cover_uneven():
    from dagsfm_amd import sharding
    for n in (0, 1, 7, 8, 9, 124750):
        for w in (1, 2, 4, 8):
            b = sharding.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1


# ---- the real assembly code (sharding.gather_match_graph, the function bench.py times) over a stub result holder

class _StubSource:
    """Stands in for a dsm_ctx: serves the results of this rank's shard as CPU tensors.  The per-pair results are a
    deterministic function of the pair, so that any rank layout must assemble to the same graph."""

    def __init__(self, pairs, verify):
        self.pairs = pairs
        c, self.rows = _fake_results(pairs)
        self.offs = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
        ic = (c // 2).astype(np.int64)
        self.ioffs = np.concatenate([[0], np.cumsum(ic)]).astype(np.int64)
        self.irows = np.concatenate([self.rows[self.offs[k]:self.offs[k] + ic[k]] for k in range(len(pairs))] or
                                    [np.zeros((0, 2), np.int32)]).astype(np.int32).reshape(-1, 2)
        rec = np.zeros((len(pairs), 328), dtype=np.uint8)  # sizeof(dsm_two_view_geometry)
        rec[:, 0] = (pairs[:, 0] + pairs[:, 1]) % 9
        rec[:, 4] = ic % 256
        rec[:, 327] = pairs[:, 1] % 251
        self.rec = rec

    def match_offsets(self):
        return torch.from_numpy(self.offs)

    def matches(self, total):
        assert total == len(self.rows)
        return torch.from_numpy(self.rows)

    def two_view_geometries(self):
        return torch.from_numpy(self.rec)

    def inlier_offsets(self):
        return torch.from_numpy(self.ioffs)

    def inlier_matches(self, total):
        assert total == len(self.irows)
        return torch.from_numpy(self.irows)


def _graph_to_numpy(g):
    return [None if t is None else t.numpy().copy() for t in (g.match_counts, g.matches, g.tvg, g.inlier_counts, g.inlier_matches)]


def _graph_worker(rank, world, port, n_images, q):
    from dagsfm_amd import sharding, synthetic
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pairs = synthetic.exhaustive_pairs(n_images).astype(np.int64)
    bounds = sharding.shard_bounds(len(pairs), world)
    src = _StubSource(sharding.shard(pairs, rank, world), True)
    g = sharding.gather_match_graph(dist, src, rank, world, bounds, True)
    q.put((rank, _graph_to_numpy(g)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_match_graph_two_ranks_equals_single_process():
    from dagsfm_amd import sharding, synthetic
    for n_images in (9, 4):  # 36 pairs (even split) and 6 pairs
        pairs = synthetic.exhaustive_pairs(n_images).astype(np.int64)
        ref = _graph_to_numpy(sharding.gather_match_graph(None, _StubSource(pairs, True), 0, 1, sharding.shard_bounds(len(pairs), 1), True))
        assert ref[0].sum() == len(ref[1]) and ref[3].sum() == len(ref[4]) and ref[2].shape == (len(pairs), 328)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_graph_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, got in res:
            for a, b in zip(got, ref):
                assert a.shape == b.shape and (a == b).all(), rank


@pytest.mark.parametrize("world,n_images", [(4, 3), (8, 4), (4, 9)])
def test_gather_match_graph_more_ranks_and_empty_shards(world, n_images):
    """4 and 8 ranks (the driver's scaling runs), including more ranks than pairs: a rank with an empty shard takes part in
    every collective with zero rows, and the assembled graph still equals the single-process one."""
    from dagsfm_amd import sharding, synthetic
    pairs = synthetic.exhaustive_pairs(n_images).astype(np.int64)
    ref = _graph_to_numpy(sharding.gather_match_graph(None, _StubSource(pairs, True), 0, 1, sharding.shard_bounds(len(pairs), 1), True))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world))
    for rank, got in res:
        for a, b in zip(got, ref):
            assert a.shape == b.shape and (a == b).all(), rank


def _single_rank_forced(port, q):
    from dagsfm_amd import sharding, synthetic
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    pairs = synthetic.exhaustive_pairs(7).astype(np.int64)
    bounds = sharding.shard_bounds(len(pairs), 1)
    out = [_graph_to_numpy(sharding.gather_match_graph(dist, _StubSource(pairs, True), 0, 1, bounds, True, force_collectives=f))
           for f in (None, True, "padded", "broadcast")]
    q.put(out)
    dist.destroy_process_group()


def test_single_rank_forced_through_the_collectives():
    """force_collectives: a lone rank takes the same all_gather_into_tensor / broadcast calls the N-rank exchange makes (what
    tests/test_rccl_single_rank_gpu.py runs over RCCL on the one GPU of the test box); the graph must not change."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_forced, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    for got in out[1:]:
        for a, b in zip(got, out[0]):
            assert a.shape == b.shape and (a == b).all()


# ---- the interleaved cut (round 6): blocks of the list dealt out to the ranks, results put back into list order

def test_interleaved_parts_partition_and_balance():
    from dagsfm_amd import sharding
    for n in (0, 1, 5, 256, 257, 1000, 124750):
        for w in (1, 2, 4, 8):
            parts = sharding.interleaved_parts(n, w, block=64)
            allp = np.concatenate(parts) if n else np.zeros(0, np.int64)
            assert sorted(allp.tolist()) == list(range(n))                      # a partition of the list
            assert all((np.diff(p) > 0).all() for p in parts)                   # every share ascending
            sizes = np.array([len(p) for p in parts])
            assert sizes.max() - sizes.min() <= 64                              # equal blocks: at most one block apart
            b, order = sharding.parts_bounds_and_order(parts)
            assert b[0] == 0 and b[-1] == n and (order == allp).all()
    # every rank sees the whole list: with 8 ranks and blocks of 256 the first and the last eighth of config 2's list are in every share
    parts = sharding.interleaved_parts(124750, 8)
    for p in parts:
        assert p.min() < 124750 // 8 and p.max() >= 124750 - 124750 // 8
    # unequal costs (images of different sizes): heaviest block first to the least loaded rank
    rng = np.random.default_rng(1)
    nfeat = rng.integers(500, 9000, 80)
    pairs = np.array([(i, j) for i in range(80) for j in range(i + 1, 80)])
    costs = sharding.pair_costs(pairs, nfeat)
    parts = sharding.interleaved_parts(len(pairs), 8, costs, block=32)
    per = np.array([costs[p].sum() for p in parts])
    blocks = np.add.reduceat(costs, np.arange(0, len(costs), 32))
    assert per.max() <= costs.sum() / 8 + blocks.max()                          # the LPT bound: no rank more than one block above its share
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(pairs)))


def _interleaved_graph_worker(rank, world, port, n_images, block, q):
    from dagsfm_amd import sharding, synthetic
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    pairs = synthetic.exhaustive_pairs(n_images).astype(np.int64)
    parts = sharding.interleaved_parts(len(pairs), world, block=block)
    bounds, order = sharding.parts_bounds_and_order(parts)
    out = []
    for exch in (None, "padded", "broadcast"):
        g = sharding.gather_match_graph(dist, _StubSource(pairs[parts[rank]], True), rank, world, bounds, True, force_collectives=exch, order=order)
        out.append(_graph_to_numpy(g))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images,block", [(2, 9, 4), (4, 9, 2), (8, 4, 1), (2, 12, 256)])
def test_gather_match_graph_interleaved_cut_equals_single_process(world, n_images, block):
    """The interleaved cut through the real assembly code: every rank ends with the graph of the whole list IN LIST ORDER, the
    same arrays a single process produces -- also with more ranks than blocks (empty shares) and a block longer than the list."""
    from dagsfm_amd import sharding, synthetic
    pairs = synthetic.exhaustive_pairs(n_images).astype(np.int64)
    ref = _graph_to_numpy(sharding.gather_match_graph(None, _StubSource(pairs, True), 0, 1, sharding.shard_bounds(len(pairs), 1), True))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_interleaved_graph_worker, args=(r, world, port, n_images, block, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, outs in res:
        for got in outs:
            for a, b in zip(got, ref):
                assert a.shape == b.shape and (a == b).all(), rank
