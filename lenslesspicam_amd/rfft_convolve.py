"""
``RealFFTConvolve2D`` on the MI355X engine: drop-in for
``lensless.recon.rfft_convolve.RealFFTConvolve2D`` (rfft_convolve.py:26-223).

convolve / deconvolve = hand-written real 2-D FFT (two real rows per complex LDS transform,
four-step column passes) with the multiplication by the PSF spectrum fused between the
forward and inverse column passes; padding is done on load, ``ifftshift`` and cropping on
store, so neither costs a pass over HBM.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native
from .recon import _Boundary, _check_dtype


class RealFFTConvolve2D(_Boundary):
    def __init__(self, psf, dtype=None, pad=True, norm="ortho", rgb=None, **kwargs):
        self._init_boundary(psf, _check_dtype(dtype, isinstance(psf, torch.Tensor)))
        assert len(psf.shape) >= 4, "Expected 4D PSF of shape ([batch], depth, width, height, channels)"
        if len(psf.shape) != 4:
            raise NotImplementedError("batched PSFs (5-D) are used only by trainable models (out of scope)")
        self._use_3d = psf.shape[-4] != 1
        self._is_rgb = (psf.shape[-1] == 3) if rgb is None else rgb
        assert self._is_rgb or psf.shape[-1] == 1
        self.norm = norm
        self.dtype = self._tdtype if self.is_torch else (np.float64 if self._real == "float64" else np.float32)
        self.pad = pad
        self._engine_options = kwargs.get("engine_options", None)    # launch-plan options (include/lpc.h)
        self._handle = None
        self._handle_batch = 0
        self.set_psf(psf)

    # -- geometry helpers with the reference's names ---------------------------------------
    def _crop(self, x):
        return x[..., self._start_idx[0]:self._end_idx[0], self._start_idx[1]:self._end_idx[1], :]

    def _pad(self, v):
        if len(v.shape) == 5:
            shape = [v.shape[0]] + self._padded_shape
        elif len(v.shape) == 4:
            shape = self._padded_shape
        else:
            raise ValueError("Expected 4D or 5D tensor")
        if isinstance(v, torch.Tensor):
            vpad = torch.zeros(size=shape, dtype=v.dtype, device=v.device)
        else:
            vpad = np.zeros(shape).astype(v.dtype)
        vpad[..., self._start_idx[0]:self._end_idx[0], self._start_idx[1]:self._end_idx[1], :] = v
        return vpad

    def _make(self, batch):
        D, H, W, C = (int(s) for s in self._psf_dev.shape)
        if self._handle is not None:
            self._handle.close()
        self._handle = self._lib.create(algo=_native.ALGO_CONV, height=H, width=W, channels=C, depth=D,
                                        batch=int(batch), norm=_native.NORM[self.norm], pad=int(bool(self.pad)),
                                        options=self._engine_options)
        self._handle_batch = int(batch)
        self._handle.set_psf(self._psf_dev.data_ptr(), self._stream())

    def set_psf(self, psf):
        self._psf = psf.type(self.dtype) if isinstance(psf, torch.Tensor) else psf.astype(self.dtype)
        self._psf_dev = self._to_dev(psf)
        self._psf_shape = np.array(self._psf.shape)
        self._make(max(self._handle_batch, 1))
        h = self._handle
        self._padded_shape = [int(self._psf_shape[-4]), h.Hp, h.Wp, 3 if self._is_rgb else 1]
        self._start_idx = np.array([h.sh, h.sw])
        self._end_idx = self._start_idx + self._psf_shape[-3:-1]

    def _run(self, x, adjoint, return_fft):
        was_torch = isinstance(x, torch.Tensor)
        if len(x.shape) not in (4, 5):
            raise ValueError("Expected 4D or 5D tensor")                     # rfft_convolve.py:91
        xd = self._to_dev(x)
        lead = xd.shape[:-4]
        D, C = int(self._psf_shape[0]), int(self._psf_shape[-1])
        want = tuple(int(v) for v in (self._psf_shape[-3:-1] if self.pad else self._padded_shape[-3:-1]))
        if tuple(int(v) for v in xd.shape[-3:-1]) != want:
            raise ValueError(f"input of spatial size {tuple(xd.shape[-3:-1])}, the operator works on {want}")
        x5 = xd.reshape((-1,) + tuple(xd.shape[-4:]))
        if x5.shape[1] != D:  # broadcasting of depth, like `rfft2(x) * H`
            assert x5.shape[1] == 1, "depth of the input must be 1 or match the PSF"
            x5 = x5.expand(-1, D, -1, -1, -1)
        Cx = int(x5.shape[-1])
        split3 = False
        if Cx != C and Cx != 1:
            # three channels against a grayscale PSF: `vpad[...] = v` cannot broadcast 3 -> 1 (rfft_convolve.py:96-99),
            # while the un-padded operator's `rfft2(x) * H` does: every channel is convolved with the one PSF
            # (only the reference's torch branch broadcasts; its NumPy branch writes into a (..., 1) scratch buffer,
            # `self._padded_data[:] = x`, rfft_convolve.py:143,171, which raises for three channels)
            if self.pad or C != 1 or not self.is_torch:
                raise ValueError(f"could not broadcast an input with {Cx} channels against a PSF with {C}")
            x5 = x5.permute(0, 4, 1, 2, 3).reshape((-1,) + tuple(x5.shape[1:4]) + (1,))    # channels -> batch items
            split3 = True
        x5 = x5.contiguous()
        n = int(x5.shape[0])
        if n > self._handle_batch:
            self._make(n)
        if return_fft:      # rfft2(pad(x)) * H (or * conj(H)), natural frequency order: (n, D, Hp, Wp/2+1, C) complex
            Hp, Wp = int(self._padded_shape[1]), int(self._padded_shape[2])
            ctype = torch.complex128 if self._tdtype == torch.float64 else torch.complex64
            out = torch.empty((n, D, Hp, Wp // 2 + 1, C), dtype=ctype, device=self._device)
            self._handle.convolve_spectrum(x5.data_ptr(), out.data_ptr(), n, int(x5.shape[-1]), adjoint, self._stream())
        else:
            out = self._empty(tuple(x5.shape[:-1]) + (C,))      # a 1-channel input broadcasts over the PSF's channels
            self._handle.convolve(x5.data_ptr(), out.data_ptr(), n, int(x5.shape[-1]), adjoint, self._stream())
        if split3:
            out = out.reshape((-1, Cx) + tuple(out.shape[1:4])).permute(0, 2, 3, 4, 1).contiguous()
        out = out.reshape(tuple(lead) + tuple(out.shape[1:]))
        if was_torch:
            return out.to(x.device)
        return out.cpu().numpy()

    def convolve(self, x, return_fft=False):
        """rfft_convolve.py:133-176"""
        y = self._run(x, False, return_fft)
        assert return_fft or y.shape[-3:-1] == x.shape[-3:-1]
        return y

    def deconvolve(self, y, return_fft=False):
        """rfft_convolve.py:178-223 (multiplication by the conjugate spectrum)"""
        x = self._run(y, True, return_fft)
        assert return_fft or x.shape[-3:-1] == y.shape[-3:-1]
        return x
