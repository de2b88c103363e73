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
import time

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
    as ``psf``).  The native handle is re-created only when the local shard size changes.

    The solver writes its shard straight into the send buffer.  What comes back is a FRESH tensor the caller owns
    (``outs = [sharded(f, n) for f in batches]`` is safe), whatever the shard sizes, PSF kind or world size.
    ``reuse_output=True`` opts into the zero-copy form for steady-state pipelines: with even shards on the engine's
    device the result then IS the receive buffer -- two of them alternate, so a returned batch stays valid only until the
    call after the next one (the reference's ``apply()`` likewise returns a view of solver state, recon.py:594).
    ``gather_ms()``: duration of the most recent all-gather (HIP events on the current stream)."""

    def __init__(self, algo_cls, psf, group=None, solver=None, reuse_output=False, **algo_kwargs):
        self.group = group
        self.reuse_output = bool(reuse_output)
        self.rec = solver if solver is not None else algo_cls(psf, **algo_kwargs)
        self.psf = psf
        self.is_torch = isinstance(psf, torch.Tensor)
        self._recv = [None, None]
        self._turn = 0
        self._ev = None

    def gather_ms(self):
        """milliseconds of the last call's collective (device events; wall clock when the engine's device is the CPU)"""
        if self._ev is None:
            return None
        if isinstance(self._ev, float):
            return self._ev
        self._ev[1].synchronize()
        return float(self._ev[0].elapsed_time(self._ev[1]))

    def __call__(self, frames, n_iter):
        rec = self.rec
        world, rank = _world(self.group)
        B = int(frames.shape[0])
        lo, hi = shard_bounds(B, world, rank)
        D, H, W, C = (int(v) for v in self.psf.shape)
        dev = rec._device
        on_dev = self.is_torch and self.psf.device == dev
        if world == 1:
            rec.set_data(frames[:, None])
            return rec.apply_batch(n_iter=n_iter)
        cap = -(-B // world)                      # slot size = largest shard
        self._turn ^= 1
        if self._recv[self._turn] is None or tuple(self._recv[self._turn].shape) != (world * cap, D, H, W, C):
            self._recv[self._turn] = torch.empty((world * cap, D, H, W, C), dtype=rec._tdtype, device=dev)
        if getattr(self, "_send", None) is None or tuple(self._send.shape) != (cap, D, H, W, C):
            self._send = torch.empty((cap, D, H, W, C), dtype=rec._tdtype, device=dev)
        recv, send = self._recv[self._turn], self._send
        if hi > lo:                               # (more ranks than frames: such a rank only takes part in the gather)
            rec.set_data(frames[lo:hi][:, None])
            if on_dev and hi - lo == cap:
                rec.apply_batch(n_iter=n_iter, out=send)          # the shard lands in the send buffer
            else:
                local = rec.apply_batch(n_iter=n_iter)
                send[: hi - lo].copy_(local if self.is_torch else torch.from_numpy(local))
        if dev.type == "cuda":
            self._ev = self._ev or (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(recv, send, group=self.group)            # the single collective of the path
        if dev.type == "cuda":
            self._ev[1].record()
        else:
            self._ev = (time.perf_counter() - t0) * 1e3
        if B == world * cap:
            # even shards: the receive buffer already is the batch (handed out as it is only on request: the other
            # buffer of the pair is overwritten by the next call, this one by the call after that)
            full = recv if self.reuse_output else recv.clone()
        else:
            full = torch.cat([recv[r * cap: r * cap + (b - a)]
                              for r in range(world) for a, b in [shard_bounds(B, world, r)]], dim=0)
        if self.is_torch:
            return full.to(self.psf.device)       # (no copy when the PSF lives on the engine's device)
        return full.cpu().numpy()


def reconstruct_sharded(algo_cls, psf, frames, n_iter, group=None, solver=None, **algo_kwargs):
    """One-shot form of ``ShardedReconstructor``.  ``solver``: an existing ``algo_cls`` instance built from
    ``psf`` (skips handle creation, the PSF FFT and the workspace allocation)."""
    # (the reconstructor dies with this call, so its receive buffer can be the result)
    return ShardedReconstructor(algo_cls, psf, group=group, solver=solver, reuse_output=True, **algo_kwargs)(frames, n_iter)


class PlaneShardedReconstructor:
    """ONE frame split over the ranks by colour channel (every solver) and, for ADMM, also by depth plane
    (optional: the 3 channels of C2 over 3 GPUs, the 16 x 3 planes of C5 over 8).

    Why this is exact: no kernel of the path couples planes -- the FFTs, the TV stencil and every prox act inside one
    (depth, channel) plane; the gradient-descent family takes its step size and its default start value per channel
    ACROSS depth (gd.py:100-112), so it is split by channel only.  Each rank runs single-plane (gray) solvers for its
    units, and one ``all_gather_into_tensor`` of the finished planes closes the frame: the result equals the un-sharded
    ``algo_cls(psf).apply()`` -- bit for bit wherever both run the same launch plan (tests/test_dist.py), to float32
    round-off where the plan depends on the number of planes in flight.

    ``__call__(data, n_iter)``: ``data`` (H, W, C) or (1, H, W, C), same kind as ``psf``; returns (D, H, W, C) on
    every rank.  A rank's units are solved by as few solvers as the engine's shapes allow (handle, PSF spectrum and
    workspace built once and kept): ONE (D', H, W, C) solver when the units are whole depth planes (C5 over 8 ranks:
    6 units = 2 planes x 3 channels = one D' = 2 RGB handle), else one gray (D', H, W, 1) solver per channel.
    ``algo_kwargs`` must be scalar solver keywords (mu1, tau, n_iter, ...): array-valued ones (``initial_est``) and
    custom ``psi*`` callables describe the whole frame and raise ``ValueError``."""

    def __init__(self, algo_cls, psf, group=None, **algo_kwargs):
        from .admm import ADMM

        # every unit gets a single-plane solver built with **algo_kwargs unchanged: an array-valued keyword describes the
        # WHOLE frame (initial_est is (D,Hp,Wp,C); a custom psi / psi_gram or a per-channel schedule act on all channels)
        # and would have to be sliced per (depth, channel) -- refused up front instead of failing inside a sub-solver
        for k, v in algo_kwargs.items():
            whole = k in ("psi", "psi_adj", "psi_gram", "initial_est") or \
                (isinstance(v, (np.ndarray, torch.Tensor)) and v.ndim > 0)
            if whole and v is not None:
                raise ValueError(f"PlaneShardedReconstructor: keyword '{k}' describes the whole frame; only scalar "
                                 "solver keywords can be handed to the per-plane solvers")
        self.group = group
        self.algo_cls = algo_cls
        self.kw = algo_kwargs
        self.psf = psf
        self.is_torch = isinstance(psf, torch.Tensor)
        D, H, W, C = (int(v) for v in psf.shape)
        self.shape = (D, H, W, C)
        by_depth = issubclass(algo_cls, ADMM) and D > 1
        self.by_depth = by_depth
        # unit = (first depth plane, number of depth planes, channel)
        self.units = [(d, 1, c) for d in range(D) for c in range(C)] if by_depth else [(0, D, c) for c in range(C)]
        self._solvers = {}
        self._recv = None

    def _groups(self, lo, hi):
        """the rank's units [lo, hi) as solver groups: [(depth planes, channels, unit indices in solver-output order)]"""
        D, H, W, C = self.shape
        mine = list(range(lo, hi))
        if not mine:
            return []
        if self.units[0][1] > 1 or not self.by_depth:     # channel units that span all depth planes: one gray solver each
            return [(list(range(self.units[u][0], self.units[u][0] + self.units[u][1])), [self.units[u][2]], [u])
                    for u in mine]
        by_d = {}
        for u in mine:
            by_d.setdefault(self.units[u][0], []).append(u)
        if C > 1 and all(len(v) == C for v in by_d.values()):       # whole depth planes: one RGB solver
            ds = sorted(by_d)
            return [(ds, list(range(C)), [u for d in ds for u in sorted(by_d[d], key=lambda k: self.units[k][2])])]
        groups = []
        for c in range(C):                                          # else one gray solver per channel
            us = [u for u in mine if self.units[u][2] == c]
            if us:
                groups.append(([self.units[u][0] for u in us], [c], us))
        return groups

    def _solver(self, key, ds, cs):
        if key not in self._solvers:
            sub = self.psf[ds][:, :, :, cs] if len(cs) == 1 else self.psf[ds]
            sub = sub.contiguous() if self.is_torch else np.ascontiguousarray(sub)
            self._solvers[key] = self.algo_cls(sub, **self.kw)
        return self._solvers[key]

    def __call__(self, data, n_iter):
        world, rank = _world(self.group)
        D, H, W, C = self.shape
        if data.ndim == 4:
            assert data.shape[0] == 1, "one frame: (H, W, C) or (1, H, W, C)"
            data = data[0]
        assert tuple(data.shape) == (H, W, C), "data must match the PSF's (H, W, C)"
        lo, hi = shard_bounds(len(self.units), world, rank)
        nd = self.units[0][1]                      # depth planes per unit (the same for every unit)
        planes, dev, tdtype = {}, torch.device("cpu"), torch.float32
        for ds, cs, us in self._groups(lo, hi):
            rec = self._solver((tuple(ds), tuple(cs)), ds, cs)
            dev, tdtype = rec._device, rec._tdtype
            y = data if len(cs) == C and C > 1 else data[:, :, cs[0]:cs[0] + 1]
            rec.set_data(y.contiguous() if self.is_torch else np.ascontiguousarray(y))
            out = rec.apply(n_iter=n_iter, disp_iter=None, plot=False)          # (len(ds), H, W, len(cs))
            out = (out if self.is_torch else torch.from_numpy(out)).to(dev)
            if nd > 1:                             # channel unit over all depth planes
                planes[us[0]] = out.reshape(nd, H, W)
            else:                                  # output order: depth-major, channel-minor == the order of `us`
                flat = out.permute(0, 3, 1, 2).reshape(len(ds) * len(cs), 1, H, W) if len(cs) > 1 else out.reshape(len(ds), 1, H, W)
                for i, u in enumerate(us):
                    planes[u] = flat[i]
        planes = [planes[u] for u in range(lo, hi)]
        if world > 1:
            if not planes:                         # more ranks than units: take part in the gather only
                from . import recon                # (device / dtype of the receive buffer: dtype=None means float32)

                f64 = self.kw.get("dtype") in ("float64", torch.float64, np.float64)
                dev = recon.runtime("float64" if f64 else "float32")[1]
                tdtype = torch.float64 if f64 else torch.float32
            cap = -(-len(self.units) // world)
            if self._recv is None:
                self._recv = torch.empty((world * cap, nd, H, W), dtype=tdtype, device=dev)
                self._send = torch.zeros((cap, nd, H, W), dtype=tdtype, device=dev)
            for i, p in enumerate(planes):
                self._send[i].copy_(p)
            dist.all_gather_into_tensor(self._recv, self._send, group=self.group)    # the single collective
            allp = [self._recv[r * cap + i] for r in range(world)
                    for i in range(shard_bounds(len(self.units), world, r)[1] - shard_bounds(len(self.units), world, r)[0])]
        else:
            allp = planes
        full = torch.empty((D, H, W, C), dtype=allp[0].dtype, device=allp[0].device)
        for (d0, n, c), p in zip(self.units, allp):
            full[d0:d0 + n, :, :, c] = p
        if self.is_torch:
            return full.to(self.psf.device)
        return full.cpu().numpy()
