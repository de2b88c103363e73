"""Raw-frame preparation on the device against the reference's own outputs
(tests/golden/preprocess.npz: lensless.utils.io.load_data / load_image run on .npy inputs) and the oracle."""
import os

import numpy as np
import pytest
import torch

from lenslesspicam_amd import prep
from oracle import preprocess_oracle as po

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

CASES = {
    "a": dict(flip=True, normalize=True),
    "b": dict(flip_ud=True, gray=True, normalize=True),
    "c": dict(single_psf=True, flip_lr=True),
    "d": dict(normalize=True, dtype="float64"),
    "e": dict(normalize=True, bg_pix=None),
}


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / max(np.max(np.abs(b)), 1e-300))


def test_oracle_matches_reference_vectors():
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    for tag, kw in CASES.items():
        kw = dict(kw)
        if "dtype" in kw:
            kw["dtype"] = np.float64
        psf, data, bg = po.preprocess_pair(g[tag + "_raw_psf"], g[tag + "_raw_data"], **kw)
        assert psf.shape == g[tag + "_psf"].shape and data.shape == g[tag + "_data"].shape
        assert rel(psf, g[tag + "_psf"]) == 0.0 and rel(data, g[tag + "_data"]) == 0.0
        if g[tag + "_bg"].size:
            assert rel(bg, g[tag + "_bg"]) == 0.0
    out = po.preprocess_frame(g["f_raw"], bg=g["f_bg"], flip=True, flip_ud=True, normalize=False)
    assert rel(out, g["f_out"]) == 0.0


@pytest.mark.parametrize("tag", sorted(CASES))
def test_load_data_golden(backend, tag):
    """float32: window mean / energy are accumulated in double on the device, in float32 pairwise sums by NumPy:
    agreement to a few float32 ulps of the normalisers, not bit-exact."""
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    kw = dict(CASES[tag])
    psf, data, bg = prep.preprocess_data(g[tag + "_raw_psf"], g[tag + "_raw_data"], return_bg=True, **kw)
    tol = 1e-6 if kw.get("dtype") != "float64" else 2e-7    # "d": the reference prepares the frame in float32
    assert tuple(psf.shape) == g[tag + "_psf"].shape and tuple(data.shape) == g[tag + "_data"].shape
    assert rel(psf, g[tag + "_psf"]) <= tol
    assert rel(data, g[tag + "_data"]) <= tol
    if g[tag + "_bg"].size:
        assert rel(bg, g[tag + "_bg"]) <= tol


def test_load_image_golden_float_frame_pixel_background(backend):
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    out = prep.preprocess_frames(g["f_raw"], bg=g["f_bg"], flip=True, flip_ud=True, normalize=False)
    assert tuple(out.shape) == g["f_out"].shape
    assert rel(out, g["f_out"]) == 0.0          # pure float32 element-wise work: bit-exact


@pytest.mark.parametrize("dt,maxv", [(np.uint8, 200), (np.uint16, 1000), (np.uint16, 5000), (np.float32, 1.0)])
def test_batched_frames_match_oracle(backend, dt, maxv):
    rng = np.random.default_rng(3)
    raw = (rng.random((3, 21, 30, 3)) * maxv).astype(dt)
    bg = np.array([0.02, 0.05, 0.03], dtype=np.float32)
    if dt == np.float32:
        bg = bg * 0.5
    out = prep.preprocess_frames(raw, bg=bg, flip_lr=True, bgr_input=True, normalize=True, gray=True)
    for b in range(3):
        ref = po.preprocess_frame(raw[b], bg=bg, flip_lr=True, bgr_input=True, normalize=True)
        ref = po.rgb2gray(ref).astype(np.float32)
        assert rel(out[b:b + 1], ref) <= 2e-7
    # every frame is normalised by its OWN maximum
    plain = prep.preprocess_frames(raw, normalize=True)
    assert float(plain.reshape(3, -1).max(dim=1).values.min()) == 1.0


def test_prepared_arrays_feed_the_solver(backend):
    import lenslesspicam_amd as lpa
    from oracle import lensless_oracle as orc

    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    psf, data = prep.preprocess_data(g["a_raw_psf"], g["a_raw_data"], flip=True, normalize=True)
    rec = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
    rec.set_data(data)
    got = rec.apply(n_iter=4, disp_iter=None)
    o = orc.ADMMOracle(g["a_psf"], tau=2e-6, mu2=1e-4)
    o.set_data(g["a_data"][0])
    assert rel(got, o.apply(4)) <= 1e-5


def test_rejects_bad_arguments(backend):
    with pytest.raises(ValueError):
        prep.preprocess_frames(np.zeros((4, 4, 3), dtype=np.int32))
    from lenslesspicam_amd._native import NativeError
    with pytest.raises(NativeError):
        prep.preprocess_psf(np.ones((1, 8, 8, 3), dtype=np.uint8), bg_pix=(5, 25))   # window outside the frame
    with pytest.raises(AssertionError):
        prep.preprocess_frames(np.ones((4, 4, 3), dtype=np.uint8), bg=np.array([0.1, 0.2]))


# ------------------------------------------------------------------------------------------------- resize --
RESIZE_CASES = [((1, 40, 52, 3), dict(factor=1 / 4)), ((2, 37, 50, 3), dict(factor=1 / 4)),
                ((1, 64, 48, 1), dict(factor=0.5)), ((1, 33, 47, 3), dict(shape=(1, 20, 31, 3))),
                ((1, 20, 20, 3), dict(factor=1.5))]


def _torch_resize(x, size):
    """what image.py:59-64 does when torch + torchvision are installed: Resize(size, antialias=True) on (C, D, H, W)"""
    import torch.nn.functional as F

    t = torch.from_numpy(np.moveaxis(x, -1, 0).copy())
    r = F.interpolate(t, size=size, mode="bilinear", align_corners=False, antialias=True).numpy()
    return np.clip(np.moveaxis(r, 0, -1), x.min(), x.max())


@pytest.mark.parametrize("shape,kw", RESIZE_CASES)
def test_oracle_resize_matches_torch_antialias(shape, kw):
    rng = np.random.default_rng(3)
    for dt, tol in ((np.float32, 5e-6), (np.float64, 1e-14)):
        x = rng.random(shape).astype(dt)
        got = po.resize_aa(x, **kw)
        assert got.dtype == dt
        assert rel(got, _torch_resize(x, got.shape[-3:-1])) <= tol


@pytest.mark.parametrize("shape,kw", RESIZE_CASES)
def test_device_resize(backend, shape, kw):
    rng = np.random.default_rng(4)
    x = rng.random(shape).astype(np.float32)
    got = prep.resize(x, **kw)
    want = po.resize_aa(x, **kw)
    assert tuple(got.shape) == want.shape
    assert rel(got, want) <= 2e-6
    assert rel(got, _torch_resize(x, want.shape[-3:-1])) <= 5e-6
    x64 = rng.random(shape)
    assert rel(prep.resize(x64, **kw), po.resize_aa(x64, **kw)) <= 1e-13
    assert prep.resize(x, factor=1) is not None and tuple(prep.resize(x, factor=1).shape) == shape   # image.py:53-54


def test_load_data_with_downsample(backend, tmp_path):
    """profile/admm.py:19-26's load_data(..., downsample=4, gray=True) on .npy inputs: the oracle chains the restated
    steps in the reference's order (background, clip, resize, norm; frame: normalise, resize to the PSF, gray)."""
    rng = np.random.default_rng(6)
    raw_psf = (rng.random((1, 64, 88, 3)) ** 8 * 3500 + rng.random((1, 64, 88, 3)) * 60 + 90).astype(np.uint16)
    raw_dat = (rng.random((64, 88, 3)) * 3000 + 150).astype(np.uint16)
    pf, df = str(tmp_path / "p.npy"), str(tmp_path / "d.npy")
    np.save(pf, raw_psf)
    np.save(df, raw_dat)
    psf, data = prep.load_data(pf, df, downsample=4, gray=True, normalize=True, bgr_input=False, plot=False)
    assert psf.shape == (1, 16, 22, 1) and data.shape == (1, 16, 22, 1)
    # reference order (io.py:331-375, 505-552)
    p = raw_psf.astype(np.float32)
    bg = np.array([np.mean(p[:, 5:25, 5:25, i]) for i in range(3)], dtype=np.float32)
    p = np.clip(p - bg, 0, None)
    p = po.resize_aa(p, factor=1 / 4)
    p = p / np.linalg.norm(p.ravel())
    d = raw_dat.astype(np.float32)[None]
    d = np.clip(d - (bg / po.get_max_val(raw_psf)) * po.get_max_val(raw_dat), 0, None)
    d = d / d.max()
    d = po.resize_aa(d, shape=p.shape)
    want_psf, want_data = po.rgb2gray(p).astype(np.float32), po.rgb2gray(d).astype(np.float32)
    assert rel(psf, want_psf) <= 5e-6 and rel(data, want_data) <= 5e-6
    # shape + normalize (what scripts/recon/admm.py passes): load_image resizes the frame FIRST and divides by the
    # maximum of the RESIZED frame (io.py:176-190), so the returned frame peaks at exactly 1
    psf2, data2 = prep.load_data(pf, df, shape=(1, 24, 32, 3), normalize=True, bgr_input=False, plot=False, downsample=None)
    assert psf2.shape == (1, 24, 32, 3) and data2.shape == (1, 24, 32, 3)
    p2 = po.resize_aa(np.clip(raw_psf.astype(np.float32) - bg, 0, None), shape=(1, 24, 32, 3))
    p2 = p2 / np.linalg.norm(p2.ravel())
    d2 = np.clip(raw_dat.astype(np.float32)[None] - (bg / po.get_max_val(raw_psf)) * po.get_max_val(raw_dat), 0, None)
    d2 = po.resize_aa(d2, shape=(1, 24, 32, 3))
    d2 = d2 / d2.max()
    assert abs(float(np.asarray(data2).max()) - 1.0) <= 1e-6
    assert rel(psf2, p2.astype(np.float32)) <= 5e-6 and rel(data2, d2.astype(np.float32)) <= 5e-6
