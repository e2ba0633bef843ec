"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The path shards at IMAGE granularity (`/root/reference/p2p/test.py:114-181`: batch_size 1, no
cross-image state), so the data path needs no collective at all.  What is needed once per run is
the weight broadcast: rank 0 loads / draws the weights, the others receive the PACKED device tensors.
xGMI is point-to-point and ring collectives are per-link bound, so the ~1.7 GB of fp16 weights go
as a few large flat buckets rather than ~700 small tensors.
"""
from typing import List

import torch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """items of rank `rank`: i = rank (mod world) — contiguous in time across ranks, disjoint, complete"""
    return list(range(rank, n_items, world))


def broadcast_tensors(tensors, src: int = 0, bucket_bytes: int = 512 << 20, group=None):
    """in-place broadcast of many tensors as few flat buckets (per dtype)"""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    n_coll = 0
    for dtype, ts in by_dtype.items():
        bucket, size = [], 0
        def flush():
            nonlocal bucket, size, n_coll
            if not bucket:
                return
            flat = torch.cat([t.reshape(-1) for t in bucket])
            dist.broadcast(flat, src=src, group=group)
            n_coll += 1
            off = 0
            for t in bucket:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            bucket, size = [], 0
        for t in ts:
            nb = t.numel() * t.element_size()
            if bucket and size + nb > bucket_bytes:
                flush()
            bucket.append(t)
            size += nb
        flush()
    return n_coll
