"""
Unrolled ADMM *inference* on the MI355X engine: the camera-inversion stage of the reference's
``UnrolledADMM`` (``lensless/recon/unrolled_admm.py:20-240``) -- per-iteration step sizes
``mu1[i], mu2[i], mu3[i], tau[i]`` and batched measurements -- without the trainable parts
(autograd, pre/post-processor networks: out of scope).

The arithmetic is the ADMM kernels' own: the fused prox/update kernel takes the previous
iteration's parameters for the pending dual updates and the current ones for the prox, and the
spectral solve forms ``R_divmat[i]`` on the fly, so a schedule costs nothing per iteration.
"""
from __future__ import annotations

import numpy as np
import torch

from .admm import ADMM


class UnrolledADMM(ADMM):
    def __init__(self, psf, dtype=None, n_iter=5, mu1=1e-6, mu2=1e-5, mu3=4e-5, tau=0.0001, psi=None,
                 psi_adj=None, psi_gram=None, pad=False, norm="backward", **kwargs):
        for key in ("pre_process", "post_process", "background_network", "psf_network", "compensation"):
            if kwargs.get(key) is not None:
                raise NotImplementedError(f"{key}: learned components are outside the hot path (inference of the "
                                          "unrolled iterations only)")
        assert isinstance(psf, torch.Tensor), "UnrolledADMM takes torch tensors, like the reference"
        ones = torch.ones(n_iter, dtype=torch.float32)
        # same attribute names as the reference so that checkpoints' state_dict entries can be assigned
        self._mu1_p, self._mu2_p, self._mu3_p, self._tau_p = ones * mu1, ones * mu2, ones * mu3, ones * tau
        super().__init__(psf, dtype=dtype, mu1=mu1, mu2=mu2, mu3=mu3, tau=tau, psi=psi, psi_adj=psi_adj,
                         psi_gram=psi_gram, pad=pad, norm=norm, n_iter=n_iter, **kwargs)

    def set_parameters(self, mu1=None, mu2=None, mu3=None, tau=None):
        """Per-iteration values (length n_iter each), e.g. from a trained LeADMM checkpoint."""
        for name, val in (("_mu1_p", mu1), ("_mu2_p", mu2), ("_mu3_p", mu3), ("_tau_p", tau)):
            if val is not None:
                v = torch.as_tensor(np.asarray(val, dtype=np.float32)).flatten()
                assert v.numel() == self._n_iter, f"{name}: expected {self._n_iter} values"
                setattr(self, name, v)

    def load_state_dict(self, state, strict=False):
        """Accepts the unrolled parameters of a reference checkpoint; everything else is ignored."""
        self.set_parameters(**{k: state[f"_{k}_p"].detach().cpu().numpy()
                               for k in ("mu1", "mu2", "mu3", "tau") if f"_{k}_p" in state})

    def _push_schedule(self):
        # unrolled_admm.py:147-151: the learnt values enter through torch.abs(), as float32
        vals = [torch.abs(getattr(self, n)).to(torch.float32).cpu().numpy().astype(np.float64)
                for n in ("_mu1_p", "_mu2_p", "_mu3_p", "_tau_p")]
        self._handle.set_admm_schedule(*vals)

    def reset(self, batch_size=None):
        self._push_schedule()
        super().reset()

    def forward(self, batch, psfs=None, background=None):
        """``batch``: (B, D=1, H, W, C) measurements -> (B, D, H, W, C) estimates after exactly
        ``n_iter`` unrolled iterations (trainable_recon.py:297-405 without the learned stages)."""
        assert isinstance(batch, torch.Tensor) and len(batch.shape) == 5, "batch must be of shape (N, D, H, W, C)"
        if psfs is not None:
            self._set_psf(psfs)
        if background is not None:
            raise NotImplementedError("background subtraction networks are outside the hot path")
        self._data = batch
        self._upload_data()
        self.reset()
        self._iterate(self._n_iter)
        return self._form_image()

    def _form_image(self, out=None):
        # unrolled_admm.py:236-240 clips OUT of place (no state mutation): read the state directly
        B = self._handle_batch
        D, Hp, Wp, C = self._padded_shape
        out_arg, out = out, self._empty((B, D, Hp, Wp, C))
        self._handle.get_state("image_est", out.data_ptr(), self._stream())
        sh, sw = (int(v) for v in self._start_idx)
        H, W = int(self._psf_shape[1]), int(self._psf_shape[2])
        res = torch.clip(out[:, :, sh:sh + H, sw:sw + W, :], min=0.0).contiguous()
        if out_arg is not None:
            out_arg.copy_(res)
        return self._to_user(res)
