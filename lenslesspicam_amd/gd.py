"""
Projected gradient descent, Nesterov and FISTA on the MI355X engine.  Drop-in for
``lensless.recon.gd`` (gd.py:24-263): same class names, keywords and defaults.

Per iteration the engine runs four padded real 2-D FFT passes with pad-on-load,
crop-on-store, the ``- y`` of the residual, the momentum arithmetic and the non-negativity
projection all folded into the FFT row passes (lpc_gd_kernels.h).
"""
from __future__ import annotations

import inspect
import time

import numpy as np
import torch

from . import _native
from .recon import ReconstructionAlgorithm


class GradientDescentUpdate:
    """Gradient descent update techniques (gd.py:24-38)."""

    VANILLA = "vanilla"
    NESTEROV = "nesterov"
    FISTA = "fista"

    @staticmethod
    def all_values():
        return [v for k, v in inspect.getmembers(GradientDescentUpdate)
                if not k.startswith("_") and not callable(v)]


def non_neg(xi):
    """Non-negative projection (gd.py:41-59).  This is the projection the engine fuses."""
    if isinstance(xi, torch.Tensor):
        return torch.maximum(xi, torch.zeros_like(xi))
    return np.maximum(xi, 0)


class GradientDescent(ReconstructionAlgorithm):
    _ALGO = _native.ALGO_GD

    def __init__(self, psf, dtype=None, proj=non_neg, lip_fact=1.8, **kwargs):
        assert callable(proj)
        self._proj = proj
        self._lip_fact = lip_fact
        # Plug-and-play hook.  ``non_neg`` is fused into the update kernel; any
        # other ``proj`` callable, or an external denoiser (gd.py:89-92: it REPLACES the projection and is called
        # as denoiser(image_est, noise_level)), splits every iteration at ``self._form_image()``:
        # lpc_iterate_begin -> callable on the (B,D,H,W,C) estimate -> lpc_iterate_end.
        denoiser = kwargs.pop("denoiser", None)
        self._pnp = None
        if denoiser is not None:
            assert "network" in denoiser.keys() and "noise_level" in denoiser.keys()     # recon.py:310-312
            if not callable(denoiser["network"]):
                raise NotImplementedError(
                    f"Unsupported denoiser: {denoiser['network']!r} (pretrained networks are outside the hot "
                    "path; pass the denoising function itself as denoiser['network'])")
            self._pnp = (denoiser["network"], denoiser["noise_level"])
        self._hook = self._pnp is not None or proj is not non_neg
        super().__init__(psf, dtype, **kwargs)
        if self._pnp is not None:
            self._denoiser, self._denoiser_noise_level = self._pnp
            self._proj = self._denoiser

    def _apply_proj(self, x):
        if self._pnp is not None:                      # gd.py:136-140
            return self._proj(x, self._denoiser_noise_level)
        return self._proj(x)

    def _iterate(self, n):
        if not self._hook:
            return super()._iterate(n)
        for _ in range(int(n)):
            self._handle.iterate_begin(self._stream())
            projected = self._to_dev(self._apply_proj(self._image_est))
            assert tuple(projected.shape) == self._state_shape(), "the projection must keep the estimate's shape"
            self._handle.iterate_end(projected.data_ptr(), self._stream())

    def _form_image(self, out=None):
        if not self._hook:
            return super()._form_image(out=out)
        res = self._apply_proj(self._image_est)         # the reference projects again on read-out (gd.py:136-140)
        if out is not None:
            out.copy_(self._to_dev(res))
        return res

    def _config(self):
        return dict(lip_fact=float(self._lip_fact))

    @property
    def _alpha(self):
        out = self._empty((int(self._psf_shape[3]),))
        self._handle.get_state("alpha", out.data_ptr(), self._stream())
        return self._to_user(out)


class NesterovGradientDescent(GradientDescent):
    _ALGO = _native.ALGO_NESTEROV

    def __init__(self, psf, dtype=None, proj=non_neg, p=0, mu=0.9, **kwargs):
        self._p, self._mu = p, mu
        super().__init__(psf, dtype, proj, **kwargs)

    def _config(self):
        return dict(lip_fact=float(self._lip_fact), nesterov_mu=float(self._mu), nesterov_p=float(self._p))

    def reset(self, p=0, mu=0.9):
        # same signature AND same defaults as gd.py:178-181: a bare reset() (which is what the
        # base constructor and apply() issue) restores p=0, mu=0.9 whatever the constructor got
        self._p, self._mu = p, mu
        super().reset()
        if p != 0 or mu != 0.9:
            self._handle.set_momentum(float(p), float(mu), 0.0)

    def _after_new_handle(self):
        # a reset(p, mu) override survives a change of the batch size, like the reference's attributes do
        if self._p != 0 or self._mu != 0.9:
            self._handle.set_momentum(float(self._p), float(self._mu), 0.0)


class FISTA(GradientDescent):
    _ALGO = _native.ALGO_FISTA

    def __init__(self, psf, dtype=None, proj=non_neg, tk=1.0, **kwargs):
        self._initial_tk = tk
        super().__init__(psf, dtype, proj, **kwargs)
        self._tk = tk

    def _config(self):
        return dict(lip_fact=float(self._lip_fact), fista_tk=float(self._initial_tk))

    def reset(self, tk=None):
        super().reset()
        self._tk = tk if tk else self._initial_tk  # gd.py:227-232
        if tk:
            self._handle.set_momentum(0.0, 0.9, float(tk))

    def _after_new_handle(self):
        if getattr(self, "_tk", self._initial_tk) != self._initial_tk:     # a reset(tk) override, see above
            self._handle.set_momentum(0.0, 0.9, float(self._tk))


def apply_gradient_descent(psf_fp, data_fp, n_iter, verbose=False, proj=non_neg, **kwargs):
    """``lensless.recon.gd.apply_gradient_descent`` (gd.py:244-263), file-path form for ``.npy`` / ``.npz`` inputs
    (keywords = ``load_data``'s); arrays instead of paths are taken as prepared ``psf`` / ``data`` (keywords then go
    to the constructor), see ``apply_admm``."""
    import os

    if isinstance(psf_fp, (str, os.PathLike)):
        from .prep import load_data

        psf, data = load_data(psf_fp=psf_fp, data_fp=data_fp, plot=False, **kwargs)
        recon = GradientDescent(psf, n_iter=n_iter, proj=proj)
    else:
        psf, data = psf_fp, data_fp
        recon = GradientDescent(psf, n_iter=n_iter, proj=proj, **kwargs)
    recon.set_data(data)
    start = time.time()
    res = recon.apply(plot=False)
    if verbose:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        print(f"Reconstruction time : {time.time() - start} s")
        print(f"Reconstruction shape: {res.shape}")
    return res
