"""
ADMM with an anisotropic-TV prior and a non-negativity constraint on the MI355X engine.
Drop-in for ``lensless.recon.admm.ADMM`` (admm.py:24-338): same keywords and defaults.

One iteration = one fused spatial kernel (dual updates, soft-threshold prox, X/W updates,
TV adjoint stencil from LDS tiles) + four real 2-D FFT passes instead of the reference's six
(linearity: rfft2(r_k) = rfft2(r_spatial) + s H* rfft2(mu1 X - xi); H V comes from the same
spectrum as V).  See DESIGN.md.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import _native
from .recon import ReconstructionAlgorithm


class ADMM(ReconstructionAlgorithm):
    _ALGO = _native.ALGO_ADMM

    def __init__(self, psf, dtype=None, mu1=1e-6, mu2=1e-5, mu3=4e-5, tau=0.0001, psi=None, psi_adj=None,
                 psi_gram=None, pad=False, norm="backward", denoiser=None, **kwargs):
        self._mu1, self._mu2, self._mu3, self._tau = mu1, mu2, mu3, tau
        assert len(psf.shape) == 4, "PSF must be 4D: (depth, height, width, channels)."
        # A caller-supplied sparsifying operator (admm.py:104-120) cannot be fused into the kernels: U, eta and Psi(V)
        # then live here, in the caller's array kind, Psi / Psi^T / the soft-threshold run as the caller's code, and
        # the engine does the rest of every iteration in one call (lpc_admm_psi_step).
        self._custom_psi = None
        if psi is not None:
            assert psi_adj is not None
            assert psi_gram is not None
            assert callable(psi)
            assert callable(psi_adj)
            assert callable(psi_gram)
            self._custom_psi = (psi, psi_adj, psi_gram)
        if pad:
            raise NotImplementedError("ADMM iterates on the padded frame (pad=False), like the reference default")
        # Depth > 1: the reference raises NotImplementedError (admm.py:92-96).  The engine runs
        # D independent 2-D problems sharing the measurement;
        # a depth-COUPLED model has no oracle and is out of scope.
        kwargs.pop("reset", None)
        # Plug-and-play (admm.py:126-133): an external denoiser replaces the TV prox.  Passed as the function
        # itself -- denoiser={"network": fn, "noise_level": s, "use_dual": bool}; fn(x, noise_level) -> x on the
        # padded (B,D,Hp,Wp,C) estimate -- every iteration is then split at the U-update
        # (lpc_admm_pnp_begin / lpc_admm_pnp_end).
        self._pnp = None
        if denoiser is not None:
            assert "network" in denoiser.keys() and "noise_level" in denoiser.keys()      # recon.py:310-312
            if not callable(denoiser["network"]):
                raise NotImplementedError(
                    f"Unsupported denoiser: {denoiser['network']!r} (pretrained networks are outside the hot "
                    "path; pass the denoising function itself as denoiser['network'])")
            self._pnp = (denoiser["network"], denoiser["noise_level"], bool(denoiser.get("use_dual", False)))
        super().__init__(psf, dtype, pad=False, norm=norm, denoiser=None, reset=False, **kwargs)
        if self._pnp is not None:
            self._denoiser, self._denoiser_noise_level, self._denoiser_use_dual = self._pnp
        if self._custom_psi is not None:
            if self._pnp is not None:
                raise NotImplementedError("a denoiser replaces the prior: pass either psi or denoiser")
            self._Psi, self._PsiT = self._custom_psi[0], self._custom_psi[1]
            self._push_psi_gram()
        self.reset()

    # -- caller-supplied prior -------------------------------------------------------------------------------------
    def _push_psi_gram(self):
        """``self._PsiTPsi = psi_gram(self._padded_shape)`` (admm.py:118) -> |.| in R_divmat (admm.py:186-190)."""
        gram = self._custom_psi[2](self._padded_shape)
        gabs = torch.abs(gram if isinstance(gram, torch.Tensor) else torch.from_numpy(np.asarray(gram)))
        D, Hp, Wp, C = self._padded_shape
        # the reference only needs the gram to broadcast inside R_divmat (admm.py:186-190): (1,Hp,Wc,1) is as good as the
        # full (D,Hp,Wc,C); the engine keeps ONE plane, so a gram that differs between planes / channels is refused
        try:
            gabs = torch.broadcast_to(gabs, (D, Hp, Wp // 2 + 1, C))
        except RuntimeError:
            raise AssertionError(f"psi_gram must broadcast to the rfft2 spectrum {(D, Hp, Wp // 2 + 1, C)}, "
                                 f"got {tuple(gabs.shape)}") from None
        plane = gabs[0, :, :, 0]
        if not torch.allclose(gabs, plane[None, :, :, None].expand_as(gabs), rtol=1e-6, atol=0):
            raise NotImplementedError("psi_gram must be the same for every depth plane and channel")
        self._gabs_dev = plane.to(device=self._device, dtype=self._tdtype).contiguous()
        self._handle.set_psi_gram(self._gabs_dev.data_ptr(), self._stream())

    def _after_new_handle(self):
        if getattr(self, "_custom_psi", None) is not None:
            self._push_psi_gram()

    def _set_psf(self, psf):
        super()._set_psf(psf)                       # restores the finite-difference gram inside the engine
        if self._custom_psi is not None:
            self._push_psi_gram()
            self.reset()

    def reset(self):
        super().reset()
        if getattr(self, "_custom_psi", None) is not None:     # admm.py:163-184
            v = self._image_est
            psi_v = self._Psi(v)
            zeros = psi_v * 0
            self._U_c, self._eta_c = zeros, zeros * 1
            self._Psi_out = psi_v if float(v.max()) else zeros * 1

    @staticmethod
    def _soft_thresh(x, thresh):                    # admm.py:341-346
        if isinstance(x, torch.Tensor):
            return torch.sign(x) * torch.max(torch.abs(x) - thresh, torch.zeros_like(x))
        return np.sign(x) * np.maximum(0, np.abs(x) - thresh)

    def _iterate_custom_psi(self, n):
        B = self._handle_batch
        D, Hp, Wp, C = self._padded_shape
        for _ in range(int(n)):
            self._U_c = self._soft_thresh(self._Psi_out + self._eta_c / self._mu2, self._tau / self._mu2)   # :245-247
            t = self._to_dev(self._PsiT(self._mu2 * self._U_c - self._eta_c))                                # :279
            assert tuple(t.shape) == (B, D, Hp, Wp, C), "psi_adj must return the padded image shape"
            self._handle.admm_psi_step(t.data_ptr(), self._stream())        # X, W, image, xi, rho  (:252-300, :310-311)
            self._Psi_out = self._Psi(self._image_est)                      # :322
            self._eta_c = self._eta_c + self._mu2 * (self._Psi_out - self._U_c)                             # :308

    def _iterate(self, n):
        if self._custom_psi is not None:
            return self._iterate_custom_psi(n)
        if self._pnp is None:
            return super()._iterate(n)
        B = self._handle_batch
        D, Hp, Wp, C = self._padded_shape
        for _ in range(int(n)):
            x = self._empty((B, D, Hp, Wp, C))
            self._handle.admm_pnp_begin(self._denoiser_use_dual, x.data_ptr(), self._stream())
            u = self._to_dev(self._denoiser(self._to_user(x), self._denoiser_noise_level))       # admm.py:235-243
            assert tuple(u.shape) == (B, D, Hp, Wp, C), "the denoiser must keep the estimate's shape"
            self._handle.admm_pnp_end(self._denoiser_use_dual, u.data_ptr(), self._stream())

    def _config(self):
        return dict(mu1=float(self._mu1), mu2=float(self._mu2), mu3=float(self._mu3), tau=float(self._tau))

    # state inspection with the reference's attribute names (values after the same number of
    # iterations; materialised on demand, the engine does not store U and W)
    def _padded_state(self, name, trailing2=False):
        B = self._handle_batch
        D, Hp, Wp, C = self._padded_shape
        shape = (B, D, Hp, Wp, C, 2) if trailing2 else (B, D, Hp, Wp, C)
        out = self._empty(shape)
        self._handle.get_state(name, out.data_ptr(), self._stream())
        return self._to_user(out)

    _X = property(lambda self: self._padded_state("X"))
    _W = property(lambda self: self._padded_state("W"))
    _U = property(lambda self: self._U_c if self._custom_psi is not None
                  else self._padded_state("U", self._pnp is None))              # image-shaped with a denoiser
    _xi = property(lambda self: self._padded_state("xi"))
    _eta = property(lambda self: self._eta_c if self._custom_psi is not None
                    else self._padded_state("eta", self._pnp is None))
    _rho = property(lambda self: self._padded_state("rho"))
    _forward_out = property(lambda self: self._padded_state("forward_out"))


def apply_admm(psf_fp, data_fp, n_iter, verbose=False, **kwargs):
    """``lensless.recon.admm.apply_admm`` (admm.py:400-419): ``load_data(psf_fp, data_fp, plot=False, **kwargs)`` ->
    ``ADMM(psf, n_iter=n_iter)`` -> ``set_data`` -> timed ``apply(plot=False)``.  The two paths name ``.npy`` /
    ``.npz`` files (``prep.load_data``; the keywords are ``load_data``'s).  Additive: when arrays are passed instead
    of paths they are taken as the prepared ``psf`` / ``data`` and the keywords go to the ``ADMM`` constructor."""
    import os

    if isinstance(psf_fp, (str, os.PathLike)):
        from .prep import load_data

        psf, data = load_data(psf_fp=psf_fp, data_fp=data_fp, plot=False, **kwargs)
        recon = ADMM(psf, n_iter=n_iter)
    else:
        psf, data = psf_fp, data_fp
        recon = ADMM(psf, n_iter=n_iter, **kwargs)
    recon.set_data(data)
    start = time.time()
    res = recon.apply(plot=False)
    if verbose:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        print(f"Reconstruction time : {time.time() - start} s")
        print(f"Reconstruction shape: {res.shape}")
    return res
