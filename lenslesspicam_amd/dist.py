"""
Sharding a batch of measurements over the GPUs of one node (one process per GPU,
``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm).

Frames never couple, so there is NO collective inside the solver loop: every
rank reconstructs its block of frames with its own replica of the PSF spectrum, and ONE all-gather of
the final images closes the batch (``all_gather_into_tensor``: one flat receive buffer, every rank's
slot the size of the largest shard; uneven batches leave the tail of the short shards unused).

``ShardedReconstructor`` keeps the solver (native handle, PSF spectrum, workspace) alive between
batches; ``reconstruct_sharded`` is the one-shot form and accepts a ready solver through ``solver=``.
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


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class ShardedReconstructor:
    """``algo_cls(psf, **algo_kwargs)`` built ONCE per rank; ``__call__(frames, n_iter)`` reconstructs this
    rank's block of ``frames`` (B,H,W,C) and returns the full (B,D,H,W,C) result on every rank (same kind
    as ``psf``).  The native handle is re-created only when the local shard size changes."""

    def __init__(self, algo_cls, psf, group=None, solver=None, **algo_kwargs):
        self.group = group
        self.rec = solver if solver is not None else algo_cls(psf, **algo_kwargs)
        self.psf = psf
        self.is_torch = isinstance(psf, torch.Tensor)
        self._recv = None

    def __call__(self, frames, n_iter):
        rec = self.rec
        world, rank = _world(self.group)
        B = int(frames.shape[0])
        lo, hi = shard_bounds(B, world, rank)
        D, H, W, C = (int(v) for v in self.psf.shape)
        dev = rec._device
        if hi > lo:
            rec.set_data(frames[lo:hi][:, None])
            local = rec.apply_batch(n_iter=n_iter)
            if not self.is_torch:
                local = torch.from_numpy(local)
        else:                                     # more ranks than frames: this rank only takes part in the gather
            local = torch.empty((0, D, H, W, C), dtype=rec._tdtype)
        if world == 1:
            return local if self.is_torch else local.numpy()
        cap = -(-B // world)                      # slot size = largest shard
        if self._recv is None or tuple(self._recv.shape) != (world * cap, D, H, W, C):
            self._recv = torch.empty((world * cap, D, H, W, C), dtype=rec._tdtype, device=dev)
            self._send = torch.empty((cap, D, H, W, C), dtype=rec._tdtype, device=dev)
        send = self._send
        send[: hi - lo].copy_(local)
        dist.all_gather_into_tensor(self._recv, send, group=self.group)      # the single collective of the path
        if B == world * cap:
            full = self._recv.clone()             # even shards: the receive buffer already is the batch
        else:
            full = torch.cat([self._recv[r * cap: r * cap + (b - a)]
                              for r in range(world) for a, b in [shard_bounds(B, world, r)]], dim=0)
        if self.is_torch:
            return full.to(self.psf.device)
        return full.cpu().numpy()


def reconstruct_sharded(algo_cls, psf, frames, n_iter, group=None, solver=None, **algo_kwargs):
    """One-shot form of ``ShardedReconstructor``.  ``solver``: an existing ``algo_cls`` instance built from
    ``psf`` (skips handle creation, the PSF FFT and the workspace allocation)."""
    return ShardedReconstructor(algo_cls, psf, group=group, solver=solver, **algo_kwargs)(frames, n_iter)
