"""
Raw-frame preparation on the MI355X -- the step in front of ``set_data``.

Array-level counterparts of the reference's loaders (``lensless/utils/io.py``): the same keyword names and
arithmetic as ``load_image`` (:46-196), ``load_psf`` (:199-375) and ``load_data`` (:378-552), minus what lives in
front of the arrays (file decoding with cv2 / rawpy, Bayer demosaicing) and minus resizing (cv2.resize; neither
is available here, so neither has an oracle).  Raw camera buffers -- uint8, uint16 or float, NumPy or torch --
go in; float tensors on the device come out, ready for ``ADMM(psf)`` / ``set_data(data)``, with every
normaliser (frame maximum, bit depth, background level, PSF energy) computed on the device by
``lpc_preprocess_frames`` / ``lpc_preprocess_psf`` (include/lpc.h).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native
from . import recon as _recon


def _raw_to_dev(a, dev):
    if isinstance(a, np.ndarray):
        if a.dtype == np.uint16:       # torch has no arithmetic on uint16; the kernels only need the bytes
            a = torch.from_numpy(np.ascontiguousarray(a).view(np.int16))
            return a.to(dev).contiguous(), "uint16"
        a = torch.from_numpy(np.ascontiguousarray(a))
    name = {torch.uint8: "uint8", torch.int16: "uint16", torch.float32: "float32", torch.float64: "float64"}
    if hasattr(torch, "uint16"):
        name[torch.uint16] = "uint16"
    if a.dtype not in name:
        raise ValueError(f"unsupported raw dtype {a.dtype}: expected uint8, uint16, float32 or float64")
    return a.detach().to(dev).contiguous(), name[a.dtype]


def _cfg(raw_name, H, W, C, flip, flip_ud, flip_lr, bgr_input, **kw):
    cfg = _native.PrepConfig()
    cfg.raw_type = _native.RAW_TYPES[raw_name]
    cfg.height, cfg.width, cfg.channels = int(H), int(W), int(C)
    cfg.flip_ud = int(bool(flip) ^ bool(flip_ud))          # io.py:160-166: flip = both axes, then the single flips
    cfg.flip_lr = int(bool(flip) ^ bool(flip_lr))
    cfg.bgr_input = int(bool(bgr_input))
    for k, v in kw.items():
        setattr(cfg, k, int(v))
    return cfg


def preprocess_frames(raw, bg=None, flip=False, flip_ud=False, flip_lr=False, bgr_input=False, normalize=True,
                      gray=False, dtype="float32"):
    """``load_image(..., as_4d=True, return_float=True)`` (io.py:157-196) for raw frames already in memory.

    raw: (H,W), (H,W,C) or a batch (B,H,W,C); bg: C background levels (fractions of full scale if <= 1, like the
    value ``load_psf`` returns, else pixel units), array or device tensor.  Returns a (B,H,W,C') device tensor
    (B = 1 for a single frame, C' = 1 if ``gray``) -- the layout ``set_data`` takes.
    """
    lib, dev = _recon.runtime(dtype)
    tdt = torch.float64 if dtype == "float64" else torch.float32
    if raw.ndim == 2:
        raw = raw[None, :, :, None]
    elif raw.ndim == 3:
        raw = raw[None]
    assert raw.ndim == 4, "raw frames must be (H,W), (H,W,C) or (B,H,W,C)"
    r, name = _raw_to_dev(raw, dev)
    B, H, W, C = (int(v) for v in r.shape)
    cfg = _cfg(name, H, W, C, flip, flip_ud, flip_lr, bgr_input, normalize=normalize, gray=gray)
    bg_dev = None
    if bg is not None:
        bg_dev = torch.as_tensor(np.asarray(bg) if not isinstance(bg, torch.Tensor) else bg).to(dev, tdt).reshape(-1)
        assert bg_dev.numel() == C, "one background level per channel"
        bg_dev = bg_dev.contiguous()
    out = torch.empty((B, H, W, 1 if (gray and C == 3) else C), dtype=tdt, device=dev)
    lib.preprocess_frames(cfg, r.data_ptr(), B, bg_dev.data_ptr() if bg_dev is not None else None, out.data_ptr(),
                          _recon._stream_handle(dev))
    return out


def preprocess_psf(raw, bg_pix=(5, 25), flip=False, flip_ud=False, flip_lr=False, bgr_input=False,
                   single_psf=False, gray=False, out_channels=None, return_bg=False, dtype="float32"):
    """``load_psf(..., return_float=True)`` (io.py:283-375) for a raw PSF already in memory: background level from
    the corner window ``bg_pix``, clip, optional ``single_psf``, division by the l2 norm.  raw: (H,W), (H,W,C) or
    a depth stack (D,H,W,C).  Returns the (D,H,W,C') device tensor and, with ``return_bg``, the C background levels
    as fractions of full scale (device tensor) -- what ``preprocess_frames(bg=...)`` expects."""
    lib, dev = _recon.runtime(dtype)
    tdt = torch.float64 if dtype == "float64" else torch.float32
    if raw.ndim == 2:
        raw = raw[None, :, :, None]
    elif raw.ndim == 3:
        raw = raw[None]
    assert raw.ndim == 4, "raw PSF must be (H,W), (H,W,C) or (D,H,W,C)"
    r, name = _raw_to_dev(raw, dev)
    D, H, W, C = (int(v) for v in r.shape)
    single = bool(single_psf) and C == 3
    rep = (out_channels or 1) if single else C
    p0, p1 = (0, 0) if bg_pix is None else (int(bg_pix[0]), int(bg_pix[1]))
    cfg = _cfg(name, H, W, C, flip, flip_ud, flip_lr, bgr_input, single_psf=single, out_channels=rep, gray=gray,
               bg_pix0=p0, bg_pix1=p1)
    c_out = 1 if (gray and rep == 3) else rep
    psf = torch.empty((D, H, W, c_out), dtype=tdt, device=dev)
    bg = torch.zeros((C,), dtype=tdt, device=dev)
    lib.preprocess_psf(cfg, r.data_ptr(), D, psf.data_ptr(), bg.data_ptr(), _recon._stream_handle(dev))
    return (psf, bg) if return_bg else psf


def preprocess_data(raw_psf, raw_data, bg_pix=(5, 25), flip=False, flip_ud=False, flip_lr=False, gray=False,
                    single_psf=False, normalize=False, bgr_input=False, dtype=None, return_bg=False,
                    flip_psf=False):
    """``load_data`` (io.py:462-552) on arrays: the PSF's background level (as a fraction of full scale) is
    re-scaled to the frame's bit depth and removed from the frame; both come back as float device tensors,
    psf (D,H,W,C') and data (1,H,W,C') [or (B,H,W,C') for a batch of frames].

    ``flip_psf``: ``load_data`` hands ``flip`` / ``flip_ud`` / ``flip_lr`` (and ``bgr_input``) to ``load_psf`` too
    (io.py:487-489), which applies them only when the PSF is an IMAGE file (through ``load_image``, io.py:293-304) and
    ignores them for ``.npy`` / ``.npz`` stacks (io.py:272-291).  False (default) is the ``.npy`` branch the golden
    vectors cover; pass True for a raw PSF that came out of an image file, so that PSF and frame keep the same
    orientation."""
    dtype = dtype or "float32"
    if dtype not in ("float32", "float64"):
        raise ValueError("dtype must be float32 or float64")
    c_data = 1 if raw_data.ndim == 2 else int(raw_data.shape[-1])
    pf = dict(flip=flip, flip_ud=flip_ud, flip_lr=flip_lr, bgr_input=bgr_input) if flip_psf else {}
    psf, bg = preprocess_psf(raw_psf, bg_pix=bg_pix, single_psf=single_psf, gray=gray, out_channels=c_data,
                             return_bg=True, dtype=dtype, **pf)
    data = preprocess_frames(raw_data, bg=bg if bg_pix is not None else None, flip=flip, flip_ud=flip_ud,
                             flip_lr=flip_lr, bgr_input=bgr_input, normalize=normalize, gray=gray, dtype=dtype)
    if data.shape[-1] != psf.shape[-1]:
        if psf.shape[-1] == 1:       # io.py:553-559: a grayscale PSF is repeated over the frame's channels
            psf = psf.repeat(1, 1, 1, data.shape[-1])
        elif data.shape[-1] == 1:    # io.py:561-567
            data = data.repeat(1, 1, 1, psf.shape[-1])
    return (psf, data, bg) if return_bg else (psf, data)


def resize(img, factor=None, shape=None, dtype=None):
    """``lensless.utils.image.resize`` (image.py:28-80), the branch the reference takes when torch is installed:
    anti-aliased bilinear resampling (torchvision ``Resize(size, antialias=True)``), then a clip to the input's range.
    ``img``: channels-last (..., H, W, C) array or tensor; returns a device tensor of the same rank.  Runs on the
    device (``lpc_resize_aa``).  Parity: pinned against ``torch.nn.functional.interpolate(antialias=True)`` -- the
    function torchvision calls -- because torchvision / cv2 are absent here."""
    assert not (factor is None and shape is None), "Must specify either factor or shape"
    t = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
    if dtype is None:
        dtype = "float64" if t.dtype == torch.float64 else "float32"
    lib, dev = _recon.runtime(dtype)
    tdt = torch.float64 if dtype == "float64" else torch.float32
    t = t.detach().to(dev, tdt).contiguous()
    assert t.dim() >= 3, "expected (..., H, W, C)"
    H, W, C = (int(v) for v in t.shape[-3:])
    new = (int(H * factor), int(W * factor)) if shape is None else (int(shape[-3]), int(shape[-2]))   # image.py:49-50
    if new == (H, W):
        return t
    n = int(np.prod(t.shape[:-3])) if t.dim() > 3 else 1
    out = torch.empty(tuple(t.shape[:-3]) + (new[0], new[1], C), dtype=tdt, device=dev)
    lib.resize_aa(t.data_ptr(), n, H, W, C, new[0], new[1], out.data_ptr(), _recon._stream_handle(dev))
    return out


def _load_array(fp):
    """``.npy`` / ``.npz`` only (io.py:122-123, 272-291): image decoding (cv2 / rawpy) is in front of the hot path."""
    import os

    fp = os.fspath(fp)
    if fp.endswith(".npy"):
        return np.load(fp)
    if fp.endswith(".npz"):
        archive = np.load(fp)
        if len(archive.files) == 0:
            raise ValueError("No arrays in .npz archive")
        return archive[archive.files[0]]
    raise NotImplementedError(
        f"{fp}: only .npy / .npz inputs are read here; decoding image files (cv2 / rawpy in the reference's "
        "load_image) is outside the accelerated path -- decode with your loader and call preprocess_data()")


def load_data(psf_fp, data_fp, background_fp=None, return_bg=False, remove_background=True, return_float=True,
              downsample=None, bg_pix=(5, 25), plot=True, flip=False, flip_ud=False, flip_lr=False, bayer=False,
              blue_gain=None, red_gain=None, gamma=None, gray=False, dtype=None, single_psf=False, shape=None,
              use_torch=False, torch_device="cpu", normalize=False, bgr_input=True):
    """File-path form of the preparation step with ``lensless.utils.io.load_data``'s signature (io.py:388-412), for
    ``.npy`` / ``.npz`` inputs: what ``scripts/recon/admm.py:30-51`` and ``apply_admm`` (admm.py:400-403) call.
    Arithmetic on the device (``preprocess_data``); ``use_torch=True, torch_device="cuda"`` keeps PSF and frame in
    HBM, otherwise NumPy arrays come back like the reference's default.  ``plot`` / ``gamma`` are accepted and
    ignored (display only).  ``downsample`` / ``shape`` resize on the device (``resize``: the anti-aliased bilinear
    filter of the reference's torch branch).  Not supported, each with an explicit error: image files,
    ``bayer=True``, ``background_fp`` (a second capture), ``return_float=False``."""
    if shape is None:
        assert downsample is not None                                     # io.py:465-466
    if bayer or blue_gain is not None or red_gain is not None:
        raise NotImplementedError("Bayer demosaicing / colour gains are in front of the accelerated path")
    if background_fp is not None:
        raise NotImplementedError("background_fp: pass the background to apply(background=...) instead")
    if not return_float:
        raise NotImplementedError("return_float=False (integer outputs) is not a solver input")
    raw_psf, raw_data = _load_array(psf_fp), _load_array(data_fp)
    if raw_psf.ndim == 3:                       # io.py:316-319 with use_3d (a .npy PSF): a 3-D stack is (D,H,W), gray
        raw_psf = raw_psf[..., None]
    assert raw_psf.ndim == 4, "a .npy / .npz PSF is a depth stack (D,H,W[,C]) (io.py:315-321)"
    resizing = shape is not None or (downsample is not None and downsample != 1)
    # load_image(shape=...) resizes the frame BEFORE `img /= img.max()` (io.py:176-190: background, clip, resize,
    # normalise) -- anti-aliased resampling lowers the maximum, so the order is observable (ADMM is not scale-invariant:
    # tau).  Without `shape` the frame is normalised at full resolution and only then resized to the PSF (io.py:527-529).
    late_norm = normalize and shape is not None
    res = preprocess_data(raw_psf, raw_data, bg_pix=bg_pix, flip=flip, flip_ud=flip_ud, flip_lr=flip_lr,
                          gray=gray and not resizing, single_psf=single_psf, normalize=normalize and not late_norm,
                          bgr_input=bgr_input, dtype=dtype, return_bg=True)
    psf, data, bg = res
    if resizing:
        # load_psf resizes between the background removal and the normalisation (io.py:352-375); resampling is linear
        # and its weights are positive, so it commutes with the scaling, the channel sum and rgb2gray that the device
        # pass has already applied -- up to rounding: resize the prepared PSF and renormalise its l2 norm ...
        # (rgb2gray comes last in the reference, AFTER the l2 normalisation of the colour PSF, io.py:550-552: it is
        # applied by the second device pass, not the first)
        dt = dtype or "float32"
        psf = resize(psf, factor=None if shape is not None else 1 / downsample, shape=shape, dtype=dt)
        psf = preprocess_psf(psf, bg_pix=None, gray=gray, dtype=dt)
        # ... and the frame: with `shape`, resized to it and normalised afterwards (per frame, on the device); else,
        # already normalised, resized to the PSF's size (io.py:527-529)
        if late_norm:
            data = resize(data, shape=shape, dtype=dt)
            data = preprocess_frames(data, normalize=True, gray=False, dtype=dt)
        if tuple(data.shape[-3:-1]) != tuple(psf.shape[-3:-1]):
            data = resize(data, shape=tuple(psf.shape), dtype=dt)
        if gray and data.shape[-1] == 3:
            data = preprocess_frames(data, normalize=False, gray=True, dtype=dt)
    if bg_pix is None:
        bg = torch.zeros((4,), dtype=psf.dtype, device=psf.device)         # io.py:338 (np.zeros(len(psf.shape)))
    if use_torch:
        out = tuple(a.to(torch_device) for a in (psf, data, bg))
    else:
        out = tuple(a.cpu().numpy() for a in (psf, data, bg))
    return out if return_bg else out[:2]
