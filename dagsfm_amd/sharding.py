"""Pair-list sharding across the GPUs of one node and assembly of the match graph.

Every image pair is an independent unit (the reference pushes pairs as separate
jobs, /root/reference/src/feature/matching.cc:767-813), so the pair list is block-partitioned over
the ranks -- "block-scheduled across the 8 GPUs" -- with all descriptors/keypoints replicated on
every GPU.  The only exchange is the final all-gather of the per-pair results (RCCL on the GPUs:
torch.distributed backend "nccl"; "gloo" in the CPU tests), the intra-node analogue of the
reference's DatabaseInfo merge over rpclib (src/map_reduce/distributed_task_manager.h:91-100).
"""
import numpy as np


PAIR_COST_FIXED = 4096.0 * 1024.0   # the per-pair term (verification) in units of descriptor-matrix elements


def pair_costs(pairs, n_feats):
    """Cost model of a pair: N1 * N2 (the distance matrix, K1) + a fixed term for the verification -- the same cut
    the C++ host shim makes between the devices of gpu_index (host/sift_feature_matcher_impl.h, Run())."""
    nf = np.asarray(n_feats, dtype=np.float64)
    p = np.asarray(pairs).reshape(-1, 2).astype(np.int64)
    return nf[p[:, 0]] * nf[p[:, 1]] + PAIR_COST_FIXED


def shard_bounds(n_pairs, world_size, costs=None):
    """Contiguous block of the pair list per rank.  Without `costs` the blocks have equal pair counts (pairs cost the
    same at a fixed feature count); with per-pair `costs` (pair_costs) the cuts equalise the summed cost instead --
    images of different sizes, e.g. a kNN candidate list over a heterogeneous collection."""
    if costs is None or n_pairs == 0:
        return np.linspace(0, n_pairs, world_size + 1).astype(np.int64)
    cum = np.concatenate([[0.0], np.cumsum(np.asarray(costs, dtype=np.float64))])
    assert len(cum) == n_pairs + 1
    b = np.zeros(world_size + 1, dtype=np.int64)
    b[-1] = n_pairs
    for r in range(1, world_size):
        b[r] = int(np.searchsorted(cum, cum[-1] * r / world_size, side="left"))
    return np.maximum.accumulate(np.minimum(b, n_pairs))


def shard(pairs, rank, world_size, costs=None):
    b = shard_bounds(len(pairs), world_size, costs)
    return pairs[b[rank]:b[rank + 1]]


INTERLEAVE_BLOCK = 256   # pairs per block of the interleaved cut


def interleaved_parts(n_pairs, world_size, costs=None, block=None):
    """The interleaved cut: the list in blocks of `block` consecutive pairs, dealt out to the ranks -- round-robin when the
    blocks cost the same (equal feature counts), else heaviest block first to the rank with the least work so far (LPT).  A
    rank's share is then a uniform sample of the whole list instead of one contiguous stretch of it: the number of RANSAC
    trials a pair needs -- which no a-priori cost model sees -- depends on WHERE in the list the pair is (an exhaustive list
    walks the image index; neighbouring images overlap more), so contiguous shares differ in verification time by the
    share's place in the list, interleaved ones do not.  Returns, per rank, the ascending global pair indices it owns."""
    if block is None:  # 256 pairs, fewer on a short list: every rank still gets at least eight blocks from all over the list
        block = max(1, min(INTERLEAVE_BLOCK, n_pairs // (8 * world_size)))
    n_blocks = (n_pairs + block - 1) // block
    starts = np.arange(n_blocks, dtype=np.int64) * block
    sizes = np.minimum(block, n_pairs - starts)
    if costs is None:
        bc = sizes.astype(np.float64)
    else:
        cum = np.concatenate([[0.0], np.cumsum(np.asarray(costs, dtype=np.float64))])
        bc = cum[starts + sizes] - cum[starts]
    owner = np.zeros(n_blocks, dtype=np.int64)
    if n_blocks and np.all(bc[:-1] == bc[0]):
        owner = np.arange(n_blocks, dtype=np.int64) % world_size       # equal blocks (the last may be short): round-robin
    else:
        load = np.zeros(world_size)
        for b in np.argsort(-bc, kind="stable"):                       # LPT; ties keep list order, so the deal is deterministic
            r = int(np.argmin(load))
            owner[b] = r
            load[r] += bc[b]
    parts = []
    for r in range(world_size):
        mine = np.nonzero(owner == r)[0]
        parts.append(np.concatenate([np.arange(starts[b], starts[b] + sizes[b], dtype=np.int64) for b in mine])
                     if len(mine) else np.zeros(0, dtype=np.int64))
    return parts


def parts_bounds_and_order(parts):
    """Rank-major layout of an interleaved cut: bounds[r] .. bounds[r + 1] = rank r's positions, order[k] = the global pair index
    at rank-major position k (what gather_match_graph needs to put the gathered results back into list order)."""
    sizes = np.array([len(p) for p in parts], dtype=np.int64)
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    order = np.concatenate(parts) if len(parts) else np.zeros(0, dtype=np.int64)
    return bounds, order


def all_gather_fixed(dist, local, max_rows, world_size):
    """All-gather of per-pair fixed-size records: `local` [rows, k] torch tensor, rows <= max_rows.
    Returns [world_size * max_rows, k]; rank r's rows start at r * max_rows."""
    import torch
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world_size * max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return out


def all_gather_ragged(dist, local, world_size):
    """Two-phase all-gather of variable-length rows: sizes first, then max-padded payloads.
    Returns (sizes [world_size] int64 on host, gathered [world_size * max, k])."""
    import torch
    sizes = torch.zeros(world_size, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    sizes_h = sizes.cpu().numpy()
    mx = max(int(sizes_h.max()), 1)
    return sizes_h, all_gather_fixed(dist, local, mx, world_size)


# padding a rank's rows up to the longest rank's is free while the shards are balanced (one collective instead of one
# per rank); beyond this ratio of longest to mean the exact-size exchange moves fewer bytes than the padding wastes
RAGGED_PAD_LIMIT = 1.25


def all_gather_rows(dist, local, rank, world_size, exchange=None):
    """All ranks' variable-length rows concatenated in rank order, exact sizes on the wire when the shards are skewed:
    the sizes are all-gathered first; balanced shards (longest <= RAGGED_PAD_LIMIT x mean) then take ONE max-padded
    all-gather, skewed ones (a kNN candidate list cut by pair count over images of very different sizes) one broadcast
    per rank of exactly that rank's rows.  `exchange` = "padded" / "broadcast" forces one of the two (tests)."""
    import torch
    sizes = torch.zeros(world_size, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    sizes_h = sizes.cpu().numpy()
    total, mx = int(sizes_h.sum()), int(sizes_h.max())
    if total == 0:
        return local[:0]
    if exchange == "padded" or (exchange is None and mx * world_size <= RAGGED_PAD_LIMIT * total):
        return assemble_ragged(sizes_h, all_gather_fixed(dist, local, max(mx, 1), world_size))
    out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    at = 0
    for r in range(world_size):
        n = int(sizes_h[r])
        if n:
            part = out[at:at + n]
            if r == rank:
                part.copy_(local)
            dist.broadcast(part, src=r)
        at += n
    return out


def assemble_ragged(sizes, gathered):
    """Concatenates the valid rows of every rank in rank order (= pair-list order)."""
    import torch
    mx = gathered.shape[0] // len(sizes)
    return torch.cat([gathered[r * mx:r * mx + int(sizes[r])] for r in range(len(sizes))], dim=0)


class MatchGraph:
    """The assembled result of a pair list: what SiftFeatureMatcher::Match leaves in `matches` +
    `two_view_geometries` (/root/reference/src/feature/matching.cc:814-836), for ALL pairs in list order."""

    def __init__(self, match_counts, matches, tvg=None, inlier_counts=None, inlier_matches=None):
        self.match_counts = match_counts      # [P] int64
        self.matches = matches                # [sum, 2] int32
        self.tvg = tvg                        # [P, record bytes] uint8 or None (matching only)
        self.inlier_counts = inlier_counts    # [P] int64 or None
        self.inlier_matches = inlier_matches  # [sum, 2] int32 or None


def reorder_rows(counts_rm, rows_rm, pos):
    """Variable-length rows from rank-major order into list order: pair p (list order) sits at rank-major position pos[p]."""
    import torch
    off_rm = torch.cumsum(counts_rm, 0) - counts_rm
    cnt = counts_rm[pos]
    out_off = torch.cumsum(cnt, 0) - cnt
    total = int(cnt.sum().item())
    if total == 0:
        return cnt, rows_rm[:0]
    idx = torch.repeat_interleave(off_rm[pos] - out_off, cnt) + torch.arange(total, dtype=torch.int64, device=rows_rm.device)
    return cnt, rows_rm[idx]


def gather_match_graph(dist, source, rank, world_size, bounds, verify, force_collectives=None, order=None):
    """Assembles the match graph of the whole pair list on every rank from the per-rank shards.

    `order` (interleaved cut, parts_bounds_and_order): the ranks' shares are not contiguous stretches of the list; the gathered
    results arrive rank by rank and are put back into list order on the device (one index build + one gather per array).

    force_collectives ("padded" / "broadcast" / True = the exchange the sizes pick): a single rank ALSO goes through
    every collective (all_gather_into_tensor of the int64 counts, the uint8 records and the int32 rows; the per-rank
    broadcast) instead of returning its own tensors -- what a one-GPU box can run of the RCCL path (bench.py
    --force-collectives, tests/test_rccl_single_rank_gpu.py).

    `source` is this rank's result holder (bench.py / the CLI wrap a dsm_ctx in it; the CPU test uses a stub) with
      match_offsets() -> [n+1] int64        matches(total) -> [total, 2] int32
      two_view_geometries() -> [n, K] uint8
      inlier_offsets() -> [n+1] int64       inlier_matches(total) -> [total, 2] int32
    all torch tensors on the device the process group communicates from.  Fixed-size per-pair records (counts,
    TwoViewGeometry) go through one max-padded all-gather; the variable-length arrays through the two-phase
    (sizes, then payload) all-gather.  With world_size 1 nothing is communicated.  `bounds` = shard_bounds()."""
    import torch
    n_mine = int(bounds[rank + 1] - bounds[rank])
    maxp = int(np.diff(bounds).max()) if world_size > 0 else n_mine

    def counts_of(offs):
        return (offs[1:] - offs[:-1]).reshape(-1, 1)

    exchange = force_collectives if isinstance(force_collectives, str) else None
    alone = world_size == 1 and not force_collectives

    def gather_counts(offs):
        c = counts_of(offs)
        if alone:
            return c.reshape(-1)
        allc = all_gather_fixed(dist, c, maxp, world_size)
        return torch.cat([allc[r * maxp:r * maxp + int(bounds[r + 1] - bounds[r])] for r in range(world_size)]).reshape(-1)

    def gather_rows(rows):
        if alone:
            return rows
        return all_gather_rows(dist, rows, rank, world_size, exchange)

    offs = source.match_offsets()
    assert offs.shape[0] == n_mine + 1
    g = MatchGraph(gather_counts(offs), gather_rows(source.matches(int(offs[-1].item()))))
    if verify:
        tv = source.two_view_geometries()
        if not alone:
            allt = all_gather_fixed(dist, tv, maxp, world_size)
            tv = torch.cat([allt[r * maxp:r * maxp + int(bounds[r + 1] - bounds[r])] for r in range(world_size)])
        g.tvg = tv
        ioffs = source.inlier_offsets()
        g.inlier_counts = gather_counts(ioffs)
        g.inlier_matches = gather_rows(source.inlier_matches(int(ioffs[-1].item())))
    if order is not None and world_size > 1:
        pos = torch.empty(len(order), dtype=torch.int64)
        pos[torch.as_tensor(np.asarray(order, dtype=np.int64))] = torch.arange(len(order), dtype=torch.int64)
        pos = pos.to(g.match_counts.device)
        g.match_counts, g.matches = reorder_rows(g.match_counts, g.matches, pos)
        if verify:
            g.tvg = g.tvg[pos]
            g.inlier_counts, g.inlier_matches = reorder_rows(g.inlier_counts, g.inlier_matches, pos)
    return g


class CtxSource:
    """gather_match_graph source over a dsm_ctx: results are fetched device-to-device into torch tensors
    (the C-ABI getters accept device pointers), so nothing crosses PCIe on the way into RCCL."""

    def __init__(self, ctx, n_pairs, device):
        import ctypes
        from . import capi
        self.ctx, self.n, self.dev = ctx, n_pairs, device
        self.L = ctx._L
        self.tvg_bytes = ctypes.sizeof(capi.TwoViewGeometry)

    def _chk(self, rc):
        self.ctx._chk(rc)

    def match_offsets(self):
        import torch
        o = torch.empty(self.n + 1, dtype=torch.int64, device=self.dev)
        self._chk(self.L.dsm_get_matches(self.ctx._h, o.data_ptr(), None, 0))
        return o

    def matches(self, total):
        import torch
        m = torch.empty((total, 2), dtype=torch.int32, device=self.dev)
        if total:
            self._chk(self.L.dsm_get_matches(self.ctx._h, None, m.data_ptr(), total))
        return m

    def two_view_geometries(self):
        import torch
        t = torch.empty((self.n, self.tvg_bytes), dtype=torch.uint8, device=self.dev)
        if self.n:
            self._chk(self.L.dsm_get_two_view_geometries(self.ctx._h, t.data_ptr()))
        return t

    def inlier_offsets(self):
        import torch
        o = torch.empty(self.n + 1, dtype=torch.int64, device=self.dev)
        self._chk(self.L.dsm_get_inlier_matches(self.ctx._h, o.data_ptr(), None, 0))
        return o

    def inlier_matches(self, total):
        import torch
        m = torch.empty((total, 2), dtype=torch.int32, device=self.dev)
        if total:
            self._chk(self.L.dsm_get_inlier_matches(self.ctx._h, None, m.data_ptr(), total))
        return m


class MultiCtxSource:
    """Several contexts on one device, each holding a contiguous part of this rank's pair list: the parts'
    results concatenated in list order (offsets rebased)."""

    def __init__(self, ctxs, counts, device):
        self.parts = [CtxSource(c, n, device) for c, n in zip(ctxs, counts)]
        self.tvg_bytes = self.parts[0].tvg_bytes

    @staticmethod
    def _cat_offsets(offs):
        import torch
        out, base = [offs[0][:1] * 0], 0
        for o in offs:
            out.append(o[1:] + base)
            base = base + int(o[-1].item())
        return torch.cat(out)

    def match_offsets(self):
        self._mo = [p.match_offsets() for p in self.parts]
        return self._cat_offsets(self._mo)

    def matches(self, total):
        import torch
        return torch.cat([p.matches(int(o[-1].item())) for p, o in zip(self.parts, self._mo)])

    def two_view_geometries(self):
        import torch
        return torch.cat([p.two_view_geometries() for p in self.parts])

    def inlier_offsets(self):
        self._io = [p.inlier_offsets() for p in self.parts]
        return self._cat_offsets(self._io)

    def inlier_matches(self, total):
        import torch
        return torch.cat([p.inlier_matches(int(o[-1].item())) for p, o in zip(self.parts, self._io)])
