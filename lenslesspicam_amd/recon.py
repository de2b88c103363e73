"""
Host-side mirror of the reference's plugin interface for iterative reconstruction
(``lensless/recon/recon.py:179-605``): same constructor keywords, ``set_data`` /
``apply`` / ``reset`` / ``get_image_estimate`` / ``_set_psf`` / ``reconstruction_error``,
same shape conventions and exception types -- but the object owns a native handle
(``include/lpc.h``) and every iteration runs in hand-written HIP kernels on the MI355X.

PyTorch is used only as plumbing: device memory for the borrowed input/output buffers and
the current HIP stream.  There is no CPU execution path.
"""
from __future__ import annotations

import abc

import numpy as np
import torch

from . import _native


def runtime(dtype="float32"):
    """(Lib, torch.device) of the product path for ``dtype``.  Fails loudly without a HIP device."""
    lib = _native.default_lib(dtype)
    return lib, torch.device("cuda", torch.cuda.current_device())


def _stream_handle(device) -> int:
    if device.type == "cuda":
        return int(torch.cuda.current_stream(device).cuda_stream)
    return 0


def _check_dtype(dtype, is_torch):
    """Mirrors ``lensless.utils.io.get_dtype`` (io.py:645-674): a string or None."""
    if dtype is None:
        dtype = "float32"
    if is_torch and isinstance(dtype, torch.dtype):
        dtype = {torch.float32: "float32", torch.float64: "float64"}.get(dtype, dtype)
    if not is_torch and dtype in (np.float32, np.float64):
        dtype = np.dtype(dtype).name
    assert dtype == "float32" or dtype == "float64"
    return dtype


class _Boundary:
    """Shared input/output plumbing: numpy or torch in, same kind out; on the device the requested dtype."""

    def _init_boundary(self, psf, dtype="float32"):
        self.is_torch = isinstance(psf, torch.Tensor)
        self._real = dtype                                   # "float32" | "float64"
        self._tdtype = torch.float64 if dtype == "float64" else torch.float32
        self._lib, self._device = runtime(dtype)
        self._out_device = psf.device if self.is_torch else None

    def _to_dev(self, a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.detach().to(device=self._device, dtype=self._tdtype).contiguous()

    def _to_user(self, t):
        if self.is_torch:
            return t.to(self._out_device)
        return t.cpu().numpy()

    def _stream(self):
        return _stream_handle(self._device)

    def _empty(self, shape):
        return torch.empty(tuple(int(s) for s in shape), dtype=self._tdtype, device=self._device)


class ReconstructionAlgorithm(_Boundary, abc.ABC):
    """Base class; see module docstring.  Sub-classes set ``_ALGO`` and create the handle."""

    _ALGO = None

    def __init__(self, psf, dtype=None, pad=True, n_iter=100, initial_est=None, reset=True,
                 denoiser=None, **kwargs):
        super().__init__()
        assert len(psf.shape) == 4, "PSF must be 4D: (depth, height, width, channels)."
        assert psf.shape[3] == 3 or psf.shape[3] == 1, "PSF must either be rgb (3) or grayscale (1)"
        self._init_boundary(psf, _check_dtype(dtype, isinstance(psf, torch.Tensor)))
        if denoiser is not None:
            raise NotImplementedError(
                "plug-and-play denoisers are outside the fused hot path (SURVEY.md section 8f, N4)"
            )
        self._psf = psf
        self._psf_dev = self._to_dev(psf)
        self._dtype = self._tdtype if self.is_torch else (np.float64 if self._real == "float64" else np.float32)
        self._npix = int(np.prod(psf.shape))
        self._n_iter = n_iter
        self._psf_shape = np.array(psf.shape)
        self._pad = pad
        self._norm = kwargs.get("norm", "ortho")
        # launch-plan options of the native engine (include/lpc.h, lpc_config.options): a dict or "k=v,k=v" string --
        # additive to the reference's keywords, which swallow unknown ones anyway (recon.py:203-213)
        self._engine_options = kwargs.get("engine_options", None)
        self._handle = None
        self._handle_batch = None
        self._data = None
        self._data_dev = None
        self._initial_est = None
        self._denoiser = None
        # geometry (rfft_convolve.py:110-117) -- queried from a throw-away operator config
        h = self._new_handle(batch=1)
        self._padded_shape = [int(psf.shape[0]), h.Hp, h.Wp, int(psf.shape[3])]
        self._start_idx = np.array([h.sh, h.sw])
        self._end_idx = self._start_idx + self._psf_shape[1:3]
        self._handle, self._handle_batch = h, 1
        self._image_est_shape = self._psf_shape if pad else np.array(self._padded_shape)
        h.set_psf(self._psf_dev.data_ptr(), self._stream())
        if initial_est is not None:
            self._set_initial_estimate(initial_est)
        if reset:
            self.reset()

    # -- native handle ----------------------------------------------------------------
    @abc.abstractmethod
    def _config(self) -> dict:
        """algorithm-specific part of ``lpc_config``"""

    def _new_handle(self, batch):
        D, H, W, C = (int(v) for v in self._psf_dev.shape)
        return self._lib.create(algo=self._ALGO, height=H, width=W, channels=C, depth=D, batch=int(batch),
                                norm=_native.NORM[self._norm], pad=int(bool(self._pad)), options=self._engine_options,
                                **self._config())

    def _ensure_handle(self, batch):
        if self._handle_batch != batch:
            if self._handle is not None:
                self._handle.close()
            self._handle = self._new_handle(batch)
            self._handle_batch = batch
            self._handle.set_psf(self._psf_dev.data_ptr(), self._stream())
            if self._initial_est is not None:
                self._push_initial_estimate()
            self._after_new_handle()

    def _after_new_handle(self):
        """Solver-specific state a fresh handle does not know yet (momentum overrides)."""

    # -- reference API ------------------------------------------------------------------
    def reset(self):
        self._handle.reset(self._stream())

    def set_data(self, data):
        if self.is_torch:
            assert isinstance(data, torch.Tensor)
        else:
            assert isinstance(data, np.ndarray)
        assert len(data.shape) >= 3, "Data must be at least 3D: [..., width, height, channel]."
        assert np.all(self._psf_shape[-3:-1] == np.array(data.shape)[-3:-1]), "PSF and data shape mismatch"
        self._check_channels(data.shape[-1], "data")
        if len(data.shape) == 3:
            self._data = data[None, None, ...]
        elif len(data.shape) == 4:
            self._data = data[None, ...]
        else:
            self._data = data
        self._upload_data()

    def _check_channels(self, ch, what):
        """The reference broadcasts a one-channel array against an RGB PSF (``vpad[...] = v``, rfft_convolve.py:96-99;
        ``- self._data``, gd.py:129) and fails on three channels against a grayscale PSF (the same assignment cannot
        broadcast 3 -> 1).  Same rule here, checked up front: the engine reads exactly B*H*W*ch values."""
        C = int(self._psf_shape[-1])
        if int(ch) != C and int(ch) != 1:
            raise ValueError(f"{what} has {int(ch)} channels, the PSF {C}: could not broadcast (only 1 -> {C} does)")

    def _upload_data(self):
        d = self._to_dev(self._data)
        assert d.shape[1] == 1, "data must have depth 1 (the measurement is 2-D)"
        B = int(d.shape[0])
        self._ensure_handle(B)
        self._data_dev = d[:, 0].contiguous()
        self._handle.set_data(self._data_dev.data_ptr(), int(d.shape[-1]), self._stream())

    def _check_est(self, image_est):
        if self.is_torch:
            assert isinstance(image_est, torch.Tensor)
        else:
            assert isinstance(image_est, np.ndarray)
        assert len(image_est.shape) >= 4, \
            "Image estimate must be at least 4D: [..., depth, width, height, channel]."
        assert np.all(self._image_est_shape[-3:-1] == np.array(image_est.shape)[-3:-1]), \
            f"Image estimate must be of shape (..., width, height, channel): {self._image_est_shape[-3:-1]}"
        assert int(image_est.shape[-1]) == int(self._psf_shape[-1]), \
            f"Image estimate must have the PSF's {int(self._psf_shape[-1])} channel(s)"
        assert int(image_est.shape[-4]) == int(self._psf_shape[0]), \
            f"Image estimate must have the PSF's depth {int(self._psf_shape[0])}"
        return image_est[None, ...] if len(image_est.shape) == 4 else image_est

    def _set_initial_estimate(self, image_est):
        """Takes effect at the next ``reset()`` (recon.py:383-413)."""
        self._initial_est = self._check_est(image_est)
        self._push_initial_estimate()

    def _push_initial_estimate(self):
        est = self._to_dev(self._initial_est)
        B = self._handle_batch
        if est.shape[0] != B:
            est = est.expand(B, *est.shape[1:]).contiguous()
        self._initial_est_dev = est
        self._handle.set_initial_estimate(est.data_ptr(), self._stream())

    def set_image_estimate(self, image_est):
        """Warm start (recon.py:415-442): the engine keeps its state in HBM, so this is
        ``_set_initial_estimate`` followed by ``reset()``."""
        self._set_initial_estimate(image_est)
        self.reset()

    def _state_shape(self):
        D, _, _, C = (int(v) for v in self._psf_shape)
        if self._pad:
            return (self._handle_batch, D, int(self._psf_shape[1]), int(self._psf_shape[2]), C)
        return (self._handle_batch, D, self._padded_shape[1], self._padded_shape[2], C)

    @property
    def _image_est(self):
        out = self._empty(self._state_shape())
        self._handle.get_state("image_est", out.data_ptr(), self._stream())
        return self._to_user(out)

    def _form_image(self, out=None):
        D, H, W, C = (int(v) for v in self._psf_shape)
        shape = (self._handle_batch, D, H, W, C)
        if out is None:
            out = self._empty(shape)
        else:      # caller's device buffer (lenslesspicam_amd.dist: the send buffer of the all-gather): no copy afterwards
            assert out.is_contiguous() and tuple(out.shape) == shape and out.dtype == self._tdtype and \
                out.device == self._device, "out= must be a contiguous device tensor of the result's shape and dtype"
        self._handle.form_image(out.data_ptr(), self._stream())
        return self._to_user(out)

    def get_image_estimate(self):
        """Current image estimate as [Batch, Depth, Height, Width, Channels]."""
        return self._form_image()

    def _set_psf(self, psf):
        assert psf.shape[-1] == 3 or psf.shape[-1] == 1, "PSF must either be rgb (3) or grayscale (1)"
        assert self._psf.shape == psf.shape, "new PSF must have same shape as old PSF"
        assert isinstance(psf, type(self._psf)), "new PSF must have same type as old PSF"
        self._psf = psf
        self._psf_dev = self._to_dev(psf)
        self._handle.set_psf(self._psf_dev.data_ptr(), self._stream())  # implies reset()

    def _progress(self):
        return

    def _get_numpy_data(self, data):
        return data.detach().cpu().numpy() if isinstance(data, torch.Tensor) else data

    def _iterate(self, n):
        self._handle.iterate(n, self._stream())

    def apply(self, n_iter=None, disp_iter=-1, plot_pause=0.2, plot=False, save=False, gamma=None, ax=None,
              reset=True, background=None, **kwargs):
        """Runs exactly ``n_iter`` iterations (recon.py:498-605) and returns the (D,H,W,C)
        estimate -- plus ``ax`` when ``plot`` is set, like the reference."""
        assert self._data is not None, "Must set data with `set_data()`"
        assert self._data.shape[0] == 1, "Apply doesn't supports processing multiple images at once."
        return self._apply_impl(n_iter, disp_iter, plot_pause, plot, save, gamma, ax, reset, background)[0]

    def apply_batch(self, n_iter=None, reset=True, background=None, out=None):
        """Additive entry: B measurements sharing the PSF in one launch sequence.  Equals B
        independent ``apply()`` calls (frames never couple); returns (B,D,H,W,C).  ``out``: a device tensor the result is
        written into (torch solvers on the engine's device only)."""
        assert self._data is not None, "Must set data with `set_data()`"
        return self._apply_impl(n_iter, None, 0.0, False, False, None, None, reset, background, out=out)[1]

    def _apply_impl(self, n_iter, disp_iter, plot_pause, plot, save, gamma, ax, reset, background, out=None):
        if background is not None:  # recon.py:553-555 (cumulative, like the reference)
            self._data = self._data - background
            self._data[self._data < 0] = 0
            self._upload_data()
        if reset:
            self.reset()
        if n_iter is None:
            n_iter = self._n_iter
        # recon.py:563-592: with `plot` or `save` set and `disp_iter` not None, an image is formed BEFORE the loop (when
        # no `ax` is handed in) and after every iteration i with `(i + 1) % disp_iter == 0` -- Python's modulo, so the
        # default disp_iter=-1 means EVERY iteration.  Each of those `_form_image()` calls is observable for ADMM,
        # whose read-out clamps its state in place (admm.py:331-338).  Iterations between two displays run as one
        # native launch sequence.
        show = (plot or save) and disp_iter is not None
        if show:
            from .plot import plot_image  # optional dependency (matplotlib)

            if ax is None:
                ax = plot_image(self._get_numpy_data(self._form_image()[0]), gamma=gamma)
            done = 0
            for k in range(1, n_iter + 1):
                if k % disp_iter != 0:
                    continue
                self._iterate(k - done)
                done = k
                self._progress()
                ax = plot_image(self._get_numpy_data(self._form_image()[0]), ax=ax, gamma=gamma,
                                title=f"Reconstruction after iteration {k}", save=save, name=f"{k}.png",
                                pause=plot_pause if plot else None)
            self._iterate(n_iter - done)
        else:
            ax = None
            self._iterate(n_iter)
        full = self._form_image(out=out)
        final_im = full[0]
        if plot:
            from .plot import plot_image

            ax = plot_image(self._get_numpy_data(final_im), ax=ax, gamma=gamma,
                            title=f"Final reconstruction after {n_iter} iterations", save=save,
                            name=f"{n_iter}.png")
            return (final_im, ax), full
        return final_im, full

    def reconstruction_error(self, prediction=None, lensless=None, psfs=None, normalize=True):
        """``|| norm(H x) - y ||^2 / npix`` per batch item (recon.py:607-653).  One convolution and
        three device reductions through ``lpc_reconstruction_error``; nothing is copied to the host
        except the final value(s) when the solver was built from NumPy arrays."""
        if prediction is None:
            prediction = self.get_image_estimate()
        if lensless is None:
            lensless = self._data
        pred = self._to_dev(prediction)
        if pred.dim() == 4:
            pred = pred[None]
        y = self._to_dev(lensless)
        y = y.reshape((-1,) + tuple(y.shape[-3:]))                       # (B,1,H,W,C) | (H,W,C) -> (B,H,W,C)
        B = int(pred.shape[0])
        D, H, W, C = (int(v) for v in self._psf_shape)
        assert tuple(pred.shape[1:]) == (D, H, W, C), \
            f"prediction must be (..., {D}, {H}, {W}, {C}), got {tuple(pred.shape)}"
        assert tuple(y.shape[1:3]) == (H, W), "PSF and data shape mismatch"
        self._check_channels(y.shape[-1], "lensless")
        y = y.expand(-1, -1, -1, C).contiguous()                         # 1 -> C broadcast, like `- lensless`
        assert y.shape[0] == B, "prediction and lensless must have the same batch size"
        own = psfs is None and B == self._handle_batch
        if own:
            h = self._handle
        else:     # another PSF of the same shape (recon.py:631-633) or another batch size: a throw-away
            psf_dev = self._psf_dev if psfs is None else self._to_dev(psfs)      # operator handle
            assert tuple(psf_dev.shape) == tuple(self._psf_dev.shape)
            D, H, W, C = (int(v) for v in psf_dev.shape)
            h = self._lib.create(algo=_native.ALGO_CONV, height=H, width=W, channels=C, depth=D, batch=B,
                                 norm=_native.NORM[self._norm], pad=1)
            h.set_psf(psf_dev.data_ptr(), self._stream())
        out = self._empty((B,))
        h.reconstruction_error(pred.data_ptr(), y.data_ptr(), bool(normalize), out.data_ptr(), self._stream())
        if not own:
            if self._device.type == "cuda":
                torch.cuda.current_stream(self._device).synchronize()
            h.close()
        return self._to_user(out)
