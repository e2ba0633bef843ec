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


def device_tensors(obj, device=None, _seen=None, _out=None):
    """every OWNING tensor (views excluded) on `device` reachable from obj through module trees, attributes, lists, tuples
    and dicts -- the packed weights of a UNet / VAE / text encoder, in a deterministic order (the same on every rank)"""
    import torch.nn as nn
    seen = _seen if _seen is not None else set()
    out = _out if _out is not None else []
    if isinstance(obj, torch.Tensor):
        if (device is None or obj.device == device) and obj._base is None and obj.data_ptr() not in seen and obj.numel() > 0:
            seen.add(obj.data_ptr())
            out.append(obj)
        return out
    if id(obj) in seen:
        return out
    if isinstance(obj, (list, tuple)):
        seen.add(id(obj))
        for v in obj:
            device_tensors(v, device, seen, out)
    elif isinstance(obj, dict):
        seen.add(id(obj))
        for k in sorted(obj, key=str):
            device_tensors(obj[k], device, seen, out)
    elif isinstance(obj, nn.Module):
        seen.add(id(obj))
        for _, t in list(obj.named_parameters(recurse=False)) + list(obj.named_buffers(recurse=False)):
            device_tensors(t.data if isinstance(t, nn.Parameter) else t, device, seen, out)
        for k in sorted(vars(obj)):
            if not k.startswith("_parameters") and k not in ("_buffers", "_modules") and not k.startswith("_state_dict"):
                device_tensors(vars(obj)[k], device, seen, out)
        for _, m in obj.named_children():
            device_tensors(m, device, seen, out)
    elif hasattr(obj, "__dict__") and not isinstance(obj, type) and type(obj).__module__.split(".")[0] in ("ief_amd", "__main__"):
        seen.add(id(obj))
        for k in sorted(vars(obj)):
            if not k.startswith("_state_dict"):
                device_tensors(vars(obj)[k], device, seen, out)
    return out


def broadcast_pipeline(pipe, src: int = 0, group=None):
    """the ONE collective of a multi-GPU run (north_star: "RCCL broadcast of UNet weights over xGMI"): rank `src` holds the
    loaded / drawn weights, every other rank built its pipeline with `empty_weights=True`; all packed device tensors of the
    UNet, the VAE and the text encoder(s) travel as a few flat buckets.  Returns the number of broadcasts issued."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    dev = pipe.unet.device
    ts = device_tensors(pipe.unet, dev) + device_tensors(pipe.vae, dev)
    for enc in (getattr(pipe, "text_encoder", None), getattr(pipe, "text_encoder_2", None)):
        if enc is not None:
            ts += device_tensors(enc, None)
    # drop duplicates across the three walks, keep order
    seen, uniq = set(), []
    for t in ts:
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append(t)
    # lazily derived tensors must not exist yet on any rank (they would be stale copies of the zeros)
    return broadcast_tensors(uniq, src=src, group=group)
