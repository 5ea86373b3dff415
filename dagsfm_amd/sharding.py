"""Pair-list sharding across the GPUs of one node and assembly of the match graph.

Every image pair is an independent unit (the reference pushes pairs as separate
jobs, /root/reference/src/feature/matching.cc:767-813), so the pair list is block-partitioned over
the ranks -- "block-scheduled across the 8 GPUs" -- with all descriptors/keypoints replicated on
every GPU.  The only exchange is the final all-gather of the per-pair results (RCCL on the GPUs:
torch.distributed backend "nccl"; "gloo" in the CPU tests), the intra-node analogue of the
reference's DatabaseInfo merge over rpclib (src/map_reduce/distributed_task_manager.h:91-100).
"""
import numpy as np


def shard_bounds(n_pairs, world_size):
    """Contiguous block of the pair list per rank (pairs cost the same at fixed feature count)."""
    return np.linspace(0, n_pairs, world_size + 1).astype(np.int64)


def shard(pairs, rank, world_size):
    b = shard_bounds(len(pairs), world_size)
    return pairs[b[rank]:b[rank + 1]]


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


def assemble_ragged(sizes, gathered):
    """Concatenates the valid rows of every rank in rank order (= pair-list order)."""
    import torch
    mx = gathered.shape[0] // len(sizes)
    return torch.cat([gathered[r * mx:r * mx + int(sizes[r])] for r in range(len(sizes))], dim=0)
