"""
Unrolled FISTA *inference* on the MI355X engine: the iterations of the reference's ``UnrolledFISTA``
(``lensless/recon/unrolled_fista.py:18-106``) with per-iteration, per-channel steps ``alpha[i][c]`` and a
``t_k`` sequence, on batches -- without autograd and the pre/post-processor networks.
"""
from __future__ import annotations

import numpy as np
import torch

from .gd import FISTA, non_neg


class UnrolledFISTA(FISTA):
    def __init__(self, psf, n_iter=5, dtype=None, proj=non_neg, learn_tk=True, tk=1, **kwargs):
        assert isinstance(psf, torch.Tensor), "UnrolledFISTA takes torch tensors, like the reference"
        super().__init__(psf, dtype=dtype, proj=proj, tk=float(tk), n_iter=n_iter, **kwargs)
        C = int(self._psf_shape[3])
        # unrolled_fista.py:60-72: alpha initialised to 1.8 / max|H* H| per channel, for every iteration
        a0 = torch.as_tensor(np.asarray(self._alpha if not isinstance(self._alpha, torch.Tensor)
                                        else self._alpha.cpu().numpy(), dtype=np.float32))
        self._alpha_p = torch.ones(n_iter, C, dtype=torch.float32) * a0
        tks = [float(tk)]                                     # unrolled_fista.py:75-78
        for i in range(n_iter):
            tks.append((1 + np.sqrt(1 + 4 * tks[i] ** 2)) / 2)
        self._tk_p = torch.Tensor(tks)
        self._sched_dirty = True

    def set_parameters(self, alpha=None, tk=None):
        self._sched_dirty = True
        if alpha is not None:
            a = torch.as_tensor(np.asarray(alpha, dtype=np.float32))
            assert tuple(a.shape) == tuple(self._alpha_p.shape)
            self._alpha_p = a
        if tk is not None:
            t = torch.as_tensor(np.asarray(tk, dtype=np.float32)).flatten()
            assert t.numel() == self._n_iter + 1
            self._tk_p = t

    def load_state_dict(self, state, strict=False):
        self.set_parameters(alpha=state["_alpha_p"].detach().cpu().numpy() if "_alpha_p" in state else None,
                            tk=state["_tk_p"].detach().cpu().numpy() if "_tk_p" in state else None)

    def _push_schedule(self):
        """Hands the schedule to the handle -- only when the parameters changed or the handle is new (an upload
        is a blocking host -> device copy; ``forward()`` calls ``reset()`` for every batch)."""
        if getattr(self, "_pushed_to", None) is self._handle and not self._sched_dirty:
            return
        alpha = torch.abs(self._alpha_p).to(torch.float32)     # unrolled_fista.py:98-100 (positivity)
        tk = torch.abs(self._tk_p).to(torch.float32)
        coef = (tk[:-1] - 1) / tk[1:]                           # float32 arithmetic, like :104
        self._handle.set_fista_schedule(alpha.tolist(), coef.tolist(), self._stream())
        self._pushed_to, self._sched_dirty = self._handle, False

    def reset(self, tk=None, batch_size=None):
        if getattr(self, "_alpha_p", None) is not None:
            self._push_schedule()
        super().reset()

    def forward(self, batch, psfs=None, background=None):
        """``batch``: (B, D, H, W, C) -> (B, D, H, W, C) after exactly ``n_iter`` unrolled iterations."""
        assert isinstance(batch, torch.Tensor) and len(batch.shape) == 5, "batch must be of shape (N, D, H, W, C)"
        if background is not None:
            raise NotImplementedError("background subtraction networks are outside the hot path")
        if psfs is not None:
            self._set_psf(psfs)
        self._data = batch
        self._upload_data()
        self.reset()
        self._iterate(self._n_iter)
        return self._form_image()
