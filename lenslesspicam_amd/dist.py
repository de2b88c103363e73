"""
Sharding a batch of measurements over the GPUs of one node (one process per GPU,
``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm).

Frames never couple, so there is NO collective inside the solver loop: every
rank reconstructs its block of frames with its own replica of the PSF spectrum, and ONE all-gather of
the final images closes the batch.  Uneven batches are padded to the largest shard for the gather.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous block partition: the first ``n_items % world`` ranks get one extra frame."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reconstruct_sharded(algo_cls, psf, frames, n_iter, group=None, **algo_kwargs):
    """Reconstruct ``frames`` (B,H,W,C) with ``algo_cls(psf, **algo_kwargs)``, B sharded over the
    ranks of ``group``.  Returns the full (B,D,H,W,C) result on every rank (same kind as ``psf``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = int(frames.shape[0])
    lo, hi = shard_bounds(B, world, rank)
    rec = algo_cls(psf, **algo_kwargs)
    D, H, W, C = (int(v) for v in psf.shape)
    is_torch = isinstance(psf, torch.Tensor)
    if hi > lo:
        rec.set_data(frames[lo:hi][:, None])
        local = rec.apply_batch(n_iter=n_iter)
        local = local if is_torch else torch.from_numpy(local)
    else:
        local = torch.empty((0, D, H, W, C), dtype=rec._tdtype)
    if world == 1:
        return local if is_torch else local.numpy()
    dev = rec._device if rec._device.type == "cuda" else torch.device("cpu")
    cap = -(-B // world)
    buf = torch.zeros((cap, D, H, W, C), dtype=rec._tdtype, device=dev)
    buf[: hi - lo] = local.to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)      # the single collective of the path
    parts = []
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        parts.append(gathered[r][: b - a])
    full = torch.cat(parts, dim=0)
    if is_torch:
        return full.to(psf.device)
    return full.cpu().numpy()
