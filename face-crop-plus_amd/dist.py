"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` over RCCL (backend
"nccl" on ROCm) — or gloo in CPU tests.

The path shards by *independent batches* (the reference already treats file batches
as unordered, independent work: cropper.py:889-902), so there is no data-path
collective: rank r of R processes batches r, r+R, ...  The only collectives are a
one-time weight broadcast (rank 0 -> all, ~229 MB fp32 for the three networks) and
a final sum / max of face counts and elapsed time.
"""
from __future__ import annotations

import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def is_dist():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def rank_world():
    if not is_dist():
        return 0, 1
    import torch.distributed as dist
    return dist.get_rank(), dist.get_world_size()


def shard(items, rank: int | None = None, world: int | None = None):
    """Round-robin shard of a list of independent work units (file batches)."""
    if rank is None or world is None:
        rank, world = rank_world()
    return list(items[rank::world])


def broadcast_state_dict(sd: dict, device=None, src: int = 0):
    """Rank ``src``'s weights -> every rank as ONE flat fp32 broadcast (a few large
    messages suit xGMI's point-to-point links better than 456 tiny ones)."""
    import torch.distributed as dist
    keys = sorted(k for k in sd if not k.endswith("num_batches_tracked"))
    flat = torch.cat([sd[k].reshape(-1).float() for k in keys])
    if device is not None:
        flat = flat.to(device)
    if dist.get_rank() != src:
        flat.zero_()
    dist.broadcast(flat, src=src)
    flat = flat.cpu()
    out, off = dict(sd), 0
    for k in keys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].reshape(sd[k].shape).clone()
        off += n
    return out


def all_reduce_scalar(value, op: str = "sum", device=None):
    """Sum / max of a python number over all ranks (face counts, elapsed time)."""
    if not is_dist():
        return value
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t.item()
