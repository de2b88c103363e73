"""
On-device image metrics mirroring ``lensless/eval/metric.py:119-172`` (``mse``, ``psnr``): both images are
divided by their own maximum (``normalize=True``) and compared in one pass on the MI355X through
``lpc_image_metrics`` (include/lpc.h).  NumPy or torch inputs; a Python float comes back for one pair, like
the reference; ``*_batch`` variants keep the per-pair results on the device for evaluation loops
(``lensless/eval/benchmark.py:346-351``).

The reference computes these with scikit-image (``mean_squared_error``, ``peak_signal_noise_ratio`` with
``data_range=None``: 1 for non-negative float images, else 2); that package is not part of this image, so the
restated formula -- not an imported reference -- is what the tests check against.
"""
from __future__ import annotations

import numpy as np
import torch

from . import recon as _recon


def _pair(true, est, dtype):
    lib, dev = _recon.runtime(dtype)
    tdt = torch.float64 if dtype == "float64" else torch.float32

    def conv(a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.detach().to(device=dev, dtype=tdt).contiguous()

    t, e = conv(true), conv(est)
    assert t.shape == e.shape, "images must have the same shape"
    return lib, dev, t, e


def metrics_batch(true, est, normalize=True, dtype="float32"):
    """(n, 2) device tensor of (mse, psnr) for n image pairs stacked on the first axis."""
    lib, dev, t, e = _pair(true, est, dtype)
    n_items = int(t.shape[0])
    n = int(t[0].numel())
    out = torch.empty((n_items, 2), dtype=t.dtype, device=dev)
    stream = _recon._stream_handle(dev)
    lib.image_metrics(t.data_ptr(), e.data_ptr(), n, n_items, bool(normalize), out.data_ptr(), stream)
    return out


def mse(true, est, normalize=True, dtype="float32"):
    """metric.py:119-144"""
    return float(metrics_batch(_one(true), _one(est), normalize, dtype)[0, 0])


def psnr(true, est, normalize=True, dtype="float32"):
    """metric.py:147-172"""
    return float(metrics_batch(_one(true), _one(est), normalize, dtype)[0, 1])


def _one(a):
    return a[None]
