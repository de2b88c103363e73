#!/usr/bin/env python3
"""
Generate golden input/output vectors from the REAL reference (read-only mount at
/root/reference).  Runs ONLY in the build container; the GPU box never has the
reference.  Output: tests/golden/*.npz (inputs + expected outputs = data, no
reference source).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

``cv2`` is absent in the image; the reference imports it at module scope but
never uses it on this path, so it is mocked.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", MagicMock())
sys.path.insert(0, os.environ.get("LENSLESS_REFERENCE", "/root/reference"))

from lensless.recon.admm import ADMM, finite_diff, finite_diff_adj, finite_diff_gram, soft_thresh  # noqa: E402
from lensless.recon.gd import FISTA, GradientDescent, NesterovGradientDescent  # noqa: E402
from lensless.recon.rfft_convolve import RealFFTConvolve2D  # noqa: E402
from scipy.fftpack import next_fast_len  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_inputs(h, w, c, seed, d=1):
    rng = np.random.default_rng(seed)
    psf = rng.random((d, h, w, c)).astype(np.float32) ** 6
    psf /= np.linalg.norm(psf.ravel())
    data = rng.random((h, w, c)).astype(np.float32)
    data /= data.max()
    return psf, data


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def admm_case(name, h, w, c, seed, iters, dtype="float32", **kw):
    psf, data = make_inputs(h, w, c, seed)
    init = kw.pop("initial_est", None)
    background = kw.pop("background", None)
    two_stage = kw.pop("two_stage", None)
    tdt = torch.float32 if dtype == "float32" else torch.float64
    rec = ADMM(t(psf).to(tdt), dtype=dtype, **kw)
    if init is not None:
        rng = np.random.default_rng(seed + 100)
        init_arr = (rng.random([1] + [int(v) for v in rec._padded_shape]).astype(np.float32) * 0.1)
        rec._set_initial_estimate(t(init_arr.copy()).to(tdt))
    rec.set_data(t(data).to(tdt))
    out = {}
    snaps = {}
    rec.reset()
    bg = None
    if background is not None:
        bg = (np.random.default_rng(seed + 7).random((h, w, c)).astype(np.float32) * 0.1)
        rec._data = rec._data - t(bg).to(tdt)
        rec._data[rec._data < 0] = 0
    for i in range(max(iters)):
        rec._update(i)
        if (i + 1) in iters:
            snaps[i + 1] = {
                "V": rec._image_est.numpy().copy(),
                "X": rec._X.numpy().copy(),
                "U": rec._U.numpy().copy(),
                "W": rec._W.numpy().copy(),
                "xi": rec._xi.numpy().copy(),
                "eta": rec._eta.numpy().copy(),
                "rho": rec._rho.numpy().copy(),
                "HV": rec._forward_out.numpy().copy(),
            }
    final = rec._form_image()[0].numpy().copy()
    out.update(psf=psf, data=data, final=final, iters=np.array(iters), dtype=dtype,
               padded_shape=np.array([int(v) for v in rec._padded_shape]),
               params=np.array([rec._mu1, rec._mu2, rec._mu3, rec._tau], dtype=np.float64))
    if init is not None:
        out["initial_est"] = init_arr
    if bg is not None:
        out["background"] = bg
    for it, st in snaps.items():
        for k, v in st.items():
            out[f"it{it}_{k}"] = v
    if two_stage:
        # apply(n1) then apply(n2, reset=False) through the public API
        rec2 = ADMM(t(psf).to(tdt), dtype=dtype, **kw)
        rec2.set_data(t(data).to(tdt))
        rec2.apply(n_iter=two_stage[0], disp_iter=None, plot=False)
        res = rec2.apply(n_iter=two_stage[1], disp_iter=None, plot=False, reset=False)
        out["two_stage"] = np.array(two_stage)
        out["two_stage_final"] = res.numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items() if not k.startswith("it")})


def gd_case(name, cls, h, w, c, seed, iters, d=1, dtype="float32", **kw):
    psf, data = make_inputs(h, w, c, seed, d=d)
    tdt = torch.float32 if dtype == "float32" else torch.float64
    init = kw.pop("initial_est", None)
    rec = cls(t(psf).to(tdt), dtype=dtype, **kw)
    out = {}
    if init is not None:
        rng = np.random.default_rng(seed + 100)
        init_arr = rng.random((1, d, h, w, c)).astype(np.float32)
        out["initial_est"] = init_arr
        rec._set_initial_estimate(t(init_arr.copy()).to(tdt))
    rec.set_data(t(data).to(tdt))
    rec.reset()
    out["alpha"] = rec._alpha.numpy().copy()
    out["x0"] = rec._image_est.numpy().copy()
    for i in range(max(iters)):
        rec._update(i)
        if (i + 1) in iters:
            out[f"it{i + 1}_x"] = rec._image_est.numpy().copy()
    out.update(psf=psf, data=data, final=rec._form_image()[0].numpy().copy(), iters=np.array(iters),
               dtype=dtype)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def unrolled_admm_case(name, h, w, c, seed, n_iter, batch):
    """UnrolledADMM inference with DIFFERENT parameters per iteration on a batch (forward())."""
    from lensless.recon.unrolled_admm import UnrolledADMM

    psf, _ = make_inputs(h, w, c, seed)
    rng = np.random.default_rng(seed + 50)
    data = rng.random((batch, 1, h, w, c)).astype(np.float32)
    rec = UnrolledADMM(t(psf), n_iter=n_iter, mu1=1e-6, mu2=1e-4, mu3=4e-5, tau=2e-6, skip_unrolled=False)
    sched = {
        "mu1": (1e-6 * (1 + 0.5 * rng.random(n_iter))).astype(np.float32),
        "mu2": (1e-4 * (1 + 0.5 * rng.random(n_iter))).astype(np.float32),
        "mu3": (4e-5 * (1 + 0.5 * rng.random(n_iter))).astype(np.float32),
        "tau": (2e-6 * (1 + 0.5 * rng.random(n_iter))).astype(np.float32),
    }
    with torch.no_grad():
        rec._mu1_p.copy_(t(sched["mu1"]))
        rec._mu2_p.copy_(t(sched["mu2"]))
        rec._mu3_p.copy_(t(sched["mu3"]))
        rec._tau_p.copy_(t(sched["tau"]))
        out = rec.forward(t(data)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), psf=psf, data=data, out=out, n_iter=n_iter, **sched)
    print("wrote", name, out.shape)


def unrolled_fista_case(name, h, w, c, seed, n_iter, batch):
    from lensless.recon.unrolled_fista import UnrolledFISTA

    psf, _ = make_inputs(h, w, c, seed)
    rng = np.random.default_rng(seed + 50)
    data = rng.random((batch, 1, h, w, c)).astype(np.float32)
    rec = UnrolledFISTA(t(psf), n_iter=n_iter, tk=1)
    with torch.no_grad():
        alpha = (rec._alpha_p.detach().numpy() * (0.6 + 0.4 * rng.random((n_iter, c)))).astype(np.float32)
        tk = (rec._tk_p.detach().numpy() * (1 + 0.2 * rng.random(n_iter + 1))).astype(np.float32)
        rec._alpha_p.copy_(t(alpha))
        rec._tk_p.copy_(t(tk))
        out = rec.forward(t(data)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), psf=psf, data=data, out=out, n_iter=n_iter, alpha=alpha,
                        tk=tk)
    print("wrote", name, out.shape)


def recon_error_case(name):
    """ReconstructionAlgorithm.reconstruction_error (recon.py:607-653) through the public API: ADMM and FISTA,
    normalised and raw, default arguments and an explicit (prediction, lensless, psfs) triple, depth 2."""
    out = {}
    psf, data = make_inputs(24, 32, 3, 31)
    rec = ADMM(t(psf), tau=2e-6, mu2=1e-4)
    rec.set_data(t(data))
    rec.apply(n_iter=6, disp_iter=None, plot=False)
    out.update(admm_psf=psf, admm_data=data, admm_iters=np.array(6), admm_params=np.array([2e-6, 1e-4]),
               admm_err=rec.reconstruction_error().numpy().copy(),
               admm_err_raw=rec.reconstruction_error(normalize=False).numpy().copy())
    psf2, data2 = make_inputs(19, 27, 3, 32, d=2)
    fis = FISTA(t(psf2))
    fis.set_data(t(data2))
    fis.apply(n_iter=9, disp_iter=None, plot=False)
    out.update(fista_psf=psf2, fista_data=data2, fista_iters=np.array(9),
               fista_err=fis.reconstruction_error().numpy().copy(),
               fista_err_raw=fis.reconstruction_error(normalize=False).numpy().copy())
    # explicit arguments: a batch of 2 arbitrary predictions against 2 frames with another PSF
    rng = np.random.default_rng(33)
    pred = rng.random((2, 2, 19, 27, 3)).astype(np.float32)
    frames = rng.random((2, 1, 19, 27, 3)).astype(np.float32)
    psf3, _ = make_inputs(19, 27, 3, 34, d=2)
    out.update(x_pred=pred, x_frames=frames, x_psf=psf3,
               x_err=fis.reconstruction_error(prediction=t(pred), lensless=t(frames), psfs=t(psf3)).numpy().copy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: v for k, v in out.items() if k.endswith("err") or k.endswith("raw")})


def preprocess_case(name):
    """load_data / load_psf / load_image (lensless/utils/io.py) on .npy inputs: the raw-frame preparation in
    front of set_data.  No cv2 call is reached (no resize, bgr_input=False, no Bayer)."""
    import tempfile

    from lensless.utils.io import load_data, load_image, load_psf

    out = {}
    rng = np.random.default_rng(41)
    tmp = tempfile.mkdtemp()

    def raw_pair(tag, d, h, w, c, maxv, dt):
        psf = (rng.random((d, h, w, c)) ** 8 * maxv * 0.9 + rng.random((d, h, w, c)) * maxv * 0.02 + maxv * 0.03)
        dat = rng.random((h, w, c)) * maxv * 0.8 + maxv * 0.05
        psf, dat = psf.astype(dt), dat.astype(dt)
        np.save(os.path.join(tmp, tag + "_psf.npy"), psf)
        np.save(os.path.join(tmp, tag + "_dat.npy"), dat)
        out[tag + "_raw_psf"], out[tag + "_raw_data"] = psf, dat
        return os.path.join(tmp, tag + "_psf.npy"), os.path.join(tmp, tag + "_dat.npy")

    cases = {
        "a": dict(shape=(1, 40, 52, 3), maxv=3000, dt=np.uint16, kw=dict(flip=True, normalize=True)),
        "b": dict(shape=(1, 33, 47, 3), maxv=250, dt=np.uint8, kw=dict(flip_ud=True, gray=True, normalize=True)),
        "c": dict(shape=(2, 40, 52, 3), maxv=60000, dt=np.uint16, kw=dict(single_psf=True, flip_lr=True)),
        "d": dict(shape=(1, 30, 36, 3), maxv=1000, dt=np.uint16, kw=dict(normalize=True, dtype="float64")),
        "e": dict(shape=(1, 30, 36, 1), maxv=4000, dt=np.uint16, kw=dict(normalize=True, bg_pix=None)),
    }
    for tag, cs in cases.items():
        pf, df = raw_pair(tag, *cs["shape"], cs["maxv"], cs["dt"])
        kw = dict(downsample=1, plot=False, bgr_input=False, return_bg=True)
        kw.update(cs["kw"])
        if kw.get("bg_pix", 0) is None:
            kw["return_bg"] = False
            psf, data = load_data(pf, df, **kw)
            bg = np.zeros(0)
        else:
            psf, data, bg = load_data(pf, df, **kw)
        out[tag + "_psf"], out[tag + "_data"], out[tag + "_bg"] = psf, data, bg
        print(tag, psf.shape, psf.dtype, data.shape, float(data.max()), bg)
    # load_image alone: explicit background in pixel units on a float frame (no bit-depth scaling), no normalise
    fr = (rng.random((28, 34, 3)) * 5.0).astype(np.float32)
    np.save(os.path.join(tmp, "f.npy"), fr)
    bgv = np.array([0.4, 1.7, 0.9], dtype=np.float32)
    out["f_raw"], out["f_bg"] = fr, bgv
    out["f_out"] = load_image(os.path.join(tmp, "f.npy"), bg=bgv, as_4d=True, return_float=True, normalize=False,
                              flip=True, flip_ud=True, bgr_input=False)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def _shrink(x):
    """a NON-idempotent projection: shrink towards 0, then clamp (so that projecting twice is visible)"""
    return torch.clamp(x - 0.05, min=0)


def _fake_denoiser(x, noise_level):
    return torch.clamp(x, min=0) * (1.0 - noise_level / 100.0)


def hook_case(name):
    """Custom `proj=` callables (gd.py:67,136-140) on the three solvers through the public API, and the
    denoiser-as-projection wiring of gd.py:89-92 (the DruNet weights cannot be loaded here, so the three
    attributes the constructor would set are set by hand to a stand-in function)."""
    out = {}
    psf, data = make_inputs(22, 30, 3, 51)
    out.update(psf=psf, data=data, iters=np.array(7), noise_level=np.array(7.0))
    for nm, cls in (("gd", GradientDescent), ("nesterov", NesterovGradientDescent), ("fista", FISTA)):
        rec = cls(t(psf), proj=_shrink)
        rec.set_data(t(data))
        out[nm + "_final"] = rec.apply(n_iter=7, disp_iter=None, plot=False).numpy().copy()
        out[nm + "_state"] = rec._image_est.numpy().copy()
    rec = FISTA(t(psf))
    rec._denoiser = _fake_denoiser
    rec._denoiser_noise_level = 7.0
    rec._proj = rec._denoiser
    rec.set_data(t(data))
    out["pnp_final"] = rec.apply(n_iter=7, disp_iter=None, plot=False).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in out.items() if k.endswith("final")})


def _pnp_denoise(x, noise_level):
    """stand-in denoiser: a 3-tap smoothing along the width + a noise-level dependent shrink (keeps the shape)"""
    sm = 0.5 * x + 0.25 * (torch.roll(x, 1, dims=-2) + torch.roll(x, -1, dims=-2))
    return sm * (1.0 - noise_level / 200.0)


def admm_pnp_case(name):
    """ADMM's plug-and-play branch (admm.py:126-133,235-243,266-275,300-311) with a stand-in denoiser: the
    attributes the constructor would set from a denoiser dict are set by hand (the DruNet weights are a download),
    then reset() + apply() through the public API.  Both use_dual settings, plus a continuation after the
    in-place clamp of _form_image."""
    out = {}
    psf, data = make_inputs(22, 30, 3, 61)
    out.update(psf=psf, data=data, iters=np.array(6), noise_level=np.array(12.0), params=np.array([1e-4, 2e-4, 3e-4]))
    # the dual form has no data term: from the all-zero start it stays at zero, so both runs get a warm start
    init = (np.random.default_rng(62).random([1, 1, 45, 60, 3]).astype(np.float32) * 0.2)
    out["initial_est"] = init
    for dual in (False, True):
        rec = ADMM(t(psf), mu1=1e-4, mu2=2e-4, mu3=3e-4, initial_est=t(init.copy()))
        assert [int(v) for v in rec._padded_shape] == [1, 45, 60, 3]
        rec._denoiser = _pnp_denoise
        rec._denoiser_noise_level = 12.0
        rec._denoiser_use_dual = dual
        rec._proj = rec._denoiser
        rec._PsiT = lambda x: x
        rec.set_data(t(data))
        tag = "dual" if dual else "plain"
        out[tag + "_final"] = rec.apply(n_iter=6, disp_iter=None, plot=False).numpy().copy()
        for k in ("_image_est", "_U", "_X", "_W", "_xi", "_eta", "_rho"):
            out[tag + k] = getattr(rec, k).numpy().copy()
        out[tag + "_more"] = rec.apply(n_iter=3, disp_iter=None, plot=False, reset=False).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in out.items() if k.endswith("final") or k.endswith("more")})


def display_case(name):
    """apply() with `save=` set (recon.py:563-592): an image is formed before the loop and after every iteration i
    with `(i + 1) % disp_iter == 0` -- Python's modulo, so the default disp_iter=-1 displays EVERY iteration -- and
    each ADMM read-out clamps the state in place (admm.py:331-338).  Right after reset() the state still ALIASES the
    initial estimate (admm.py:154-155), so the pre-loop read-out clamps the stored initial estimate as well and every
    later reset() starts from the clamped one.  Needs matplotlib (Agg) for the reference's plot_image."""
    import tempfile

    os.environ.setdefault("MPLBACKEND", "Agg")
    out = {}
    psf, data = make_inputs(24, 32, 3, 71)
    kw = dict(tau=2e-6, mu2=1e-4)
    rec = ADMM(t(psf), **kw)
    init = ((np.random.default_rng(72).random([1] + [int(v) for v in rec._padded_shape]).astype(np.float32) - 0.5)
            * 0.4)                                                # warm start WITH negative entries
    out.update(psf=psf, data=data, initial_est=init.copy(), params=np.array([2e-6, 1e-4]))
    rec._set_initial_estimate(t(init.copy()))
    rec.set_data(t(data))
    tmp = tempfile.mkdtemp()
    out["run1_n9_dispm1"] = rec.apply(n_iter=9, save=tmp, disp_iter=-1).numpy().copy()
    out["run1_state"] = rec._image_est.numpy().copy()
    out["run1_files"] = np.array(sorted(int(f[:-4]) for f in os.listdir(tmp)))
    out["initial_est_after"] = rec._initial_est.numpy().copy()     # clamped in place by the pre-loop read-out
    out["run2_n7_disp3"] = rec.apply(n_iter=7, save=tempfile.mkdtemp(), disp_iter=3).numpy().copy()
    out["run3_n8_dispm3"] = rec.apply(n_iter=8, save=tempfile.mkdtemp(), disp_iter=-3).numpy().copy()
    out["run4_n5_none"] = rec.apply(n_iter=5, disp_iter=None).numpy().copy()
    # the same calls from a cold start (zeros): here the displays leave the trajectory alone
    rec0 = ADMM(t(psf), **kw)
    rec0.set_data(t(data))
    out["cold_n9_dispm1"] = rec0.apply(n_iter=9, save=tempfile.mkdtemp(), disp_iter=-1).numpy().copy()
    # gradient-descent family: the read-out is not in place (gd.py:136-140)
    fis = FISTA(t(psf))
    fis.set_data(t(data))
    out["fista_n6_dispm1"] = fis.apply(n_iter=6, save=tempfile.mkdtemp(), disp_iter=-1).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in out.items() if k.startswith("run") or "disp" in k})


def caller_flow_case(name):
    """The flow of scripts/recon/admm.py:22-130 on .npy inputs (no cv2 call is reached with downsample=1, no Bayer):
    load_data(...) -> ADMM(psf, **config.admm) with the FULL key set of configs/recon/defaults.yaml:63-82 -> set_data
    -> apply(disp_iter=None, save=False, gamma=None, plot=False); plus the file-path convenience wrappers
    apply_admm / apply_gradient_descent (admm.py:400-419, gd.py:244-263)."""
    import tempfile

    from lensless.recon.admm import apply_admm
    from lensless.recon.gd import apply_gradient_descent
    from lensless.utils.io import load_data

    rng = np.random.default_rng(81)
    tmp = tempfile.mkdtemp()
    h, w = 36, 48
    raw_psf = (rng.random((1, h, w, 3)) ** 8 * 3500 + rng.random((1, h, w, 3)) * 60 + 90).astype(np.uint16)
    raw_dat = (rng.random((h, w, 3)) * 3000 + 150).astype(np.uint16)
    pf, df = os.path.join(tmp, "psf.npy"), os.path.join(tmp, "dat.npy")
    np.save(pf, raw_psf)
    np.save(df, raw_dat)
    out = dict(raw_psf=raw_psf, raw_data=raw_dat)
    admm_cfg = dict(n_iter=5, mu1=1e-6, mu2=1e-5, mu3=4e-5, tau=0.0001, denoiser=None, unrolled=False,
                    checkpoint_fp=None, pre_process_model=dict(network=None, depth=2),
                    post_process_model=dict(network=None, depth=2))
    for tag, use_torch in (("np", False), ("torch", True)):
        psf, data = load_data(psf_fp=pf, data_fp=df, background_fp=None, dtype="float32", downsample=1, bayer=False,
                              blue_gain=None, red_gain=None, plot=False, flip=False, gamma=None, gray=False,
                              single_psf=False, shape=None, use_torch=use_torch, torch_device="cpu", bg_pix=[5, 25],
                              normalize=True, bgr_input=False)
        rec = ADMM(psf, **admm_cfg)
        rec.set_data(data)
        res = rec.apply(disp_iter=None, save=False, gamma=None, plot=False)
        out[f"script_{tag}_psf"] = np.asarray(psf)
        out[f"script_{tag}_data"] = np.asarray(data)
        out[f"script_{tag}_res"] = np.asarray(res)
    out["apply_admm_n4_flip_gray"] = np.asarray(apply_admm(pf, df, 4, downsample=1, flip=True, gray=True,
                                                           normalize=True, bgr_input=False))
    out["apply_gd_n6"] = np.asarray(apply_gradient_descent(pf, df, 6, downsample=1, normalize=True, bgr_input=False))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


# a caller-supplied sparsifying operator for ADMM (admm.py:44-46,104-120): two weighted circular differences, one of
# them over a distance of 2 pixels; psi_gram = rfft2 of the stencil of Psi^T Psi, like finite_diff_gram
def _psi2(x):
    return torch.stack((1.5 * (torch.roll(x, 1, dims=-3) - x), 0.5 * (torch.roll(x, 2, dims=-2) - x)), dim=len(x.shape))


def _psi2_adj(u):
    return 1.5 * (torch.roll(u[..., 0], -1, dims=-3) - u[..., 0]) + 0.5 * (torch.roll(u[..., 1], -2, dims=-2) - u[..., 1])


def _psi2_gram(shape, dtype=torch.float32):
    gram = torch.zeros([int(v) for v in shape], dtype=dtype)
    gram[0, 0, 0] = 2 * 1.5 ** 2 + 2 * 0.5 ** 2
    gram[0, 1, 0] = gram[0, -1, 0] = -(1.5 ** 2)
    gram[0, 0, 2] = gram[0, 0, -2] = -(0.5 ** 2)
    return torch.fft.rfft2(gram, dim=(-3, -2))


def custom_psi_case(name):
    """ADMM(psf, psi=..., psi_adj=..., psi_gram=...) through the public API, cold and warm start, with a continuation."""
    out = {}
    psf, data = make_inputs(22, 30, 3, 91)
    kw = dict(mu1=1e-4, mu2=2e-4, mu3=3e-4, tau=2e-6)
    out.update(psf=psf, data=data, params=np.array([1e-4, 2e-4, 3e-4, 2e-6]), iters=np.array(8))
    rec = ADMM(t(psf), psi=_psi2, psi_adj=_psi2_adj, psi_gram=_psi2_gram, **kw)
    # the operators are consistent: <Psi x, u> == <x, Psi^T u>
    x = torch.randn([1] + [int(v) for v in rec._padded_shape])
    u = torch.randn(list(x.shape) + [2])
    assert abs(float((_psi2(x) * u).sum() - (x * _psi2_adj(u)).sum())) < 1e-2
    rec.set_data(t(data))
    out["final"] = rec.apply(n_iter=8, disp_iter=None, plot=False).numpy().copy()
    for k in ("_image_est", "_U", "_X", "_W", "_xi", "_eta", "_rho"):
        out["state" + k] = getattr(rec, k).numpy().copy()
    out["more"] = rec.apply(n_iter=3, disp_iter=None, plot=False, reset=False).numpy().copy()
    init = (np.random.default_rng(92).random([1, 1, 45, 60, 3]).astype(np.float32) - 0.3) * 0.2
    out["initial_est"] = init
    rec2 = ADMM(t(psf), psi=_psi2, psi_adj=_psi2_adj, psi_gram=_psi2_gram, initial_est=t(init.copy()), **kw)
    rec2.set_data(t(data))
    out["warm_final"] = rec2.apply(n_iter=6, disp_iter=None, plot=False).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in out.items() if "final" in k or k == "more"},
          float(np.count_nonzero(out["state_U"]) / out["state_U"].size))



# ---- every legal `norm` and PSF scale (rfft_convolve.py:27,121; admm.py:50,101; recon.py:203-329 never normalises) ----
LADDER = (
    # tag, norm, PSF scale rule
    ("backward_l2", "backward", "l2"),        # the loaders' convention (io.py:375): unit l2 norm
    ("ortho_l2", "ortho", "l2"),
    ("forward_l2", "forward", "l2"),
    ("backward_l2_1em3", "backward", "l2*1e-3"),
    ("backward_max1", "backward", "max"),     # psf / psf.max()
    ("backward_0_255", "backward", "255"),    # an 8-bit PSF left as it came
)


def scale_psf(psf, rule):
    psf = psf.astype(np.float64)
    psf = psf / np.linalg.norm(psf.ravel())
    if rule == "l2*1e-3":
        psf = psf * 1e-3
    elif rule == "max":
        psf = psf / psf.max()
    elif rule == "255":
        psf = np.round(psf / psf.max() * 255.0)
    return psf.astype(np.float32)


def norm_scale_case(name, shapes=((36, 52, 3), (37, 53, 3))):
    """ADMM (defaults and TV-active), FISTA and the bare operator for every legal `norm` and a ladder of PSF scales.
    Stored per entry: the FLOAT64 run (the truth a float32 implementation is measured against; sensor-window crop of
    the padded ADMM estimate, taken by slicing -- no in-place clamp) and, as scalars, how far the reference's own
    float32 run is from it (max |difference| / max |truth|): the yardstick of tests/test_norm_scale.py.  The second
    (odd-sized) shape carries a shorter ladder to keep the fixture small."""
    ladders = (LADDER, tuple(l for l in LADDER if l[0] in ("backward_l2", "forward_l2", "backward_max1")))
    out = {"admm_iters": np.array([6, 20]), "fista_iters": np.array([40]), "tv_params": np.array([2e-6, 1e-4])}

    def dist(a, b):
        return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())

    for si, (h, w, c) in enumerate(shapes):
        rng = np.random.default_rng(300 + si)
        base = rng.random((1, h, w, c)) ** 6
        data = rng.random((h, w, c)).astype(np.float32)
        data /= data.max()
        x = rng.random((1, 1, h, w, c)).astype(np.float32)
        out[f"s{si}_shape"] = np.array([h, w, c])
        out[f"s{si}_data"], out[f"s{si}_x"] = data, x
        out[f"s{si}_tags"] = np.array([l[0] for l in ladders[si]])
        out[f"s{si}_norms"] = np.array([l[1] for l in ladders[si]])
        for tag, norm, rule in ladders[si]:
            psf = scale_psf(base, rule)
            out[f"s{si}_{tag}_psf"] = psf
            res = {}
            for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
                dtype = "float32" if dt == "f32" else "float64"
                r = res.setdefault(dt, {})
                cv = RealFFTConvolve2D(t(psf).to(tdt), dtype=tdt, pad=True, norm=norm)
                r["conv"] = cv.convolve(t(x).to(tdt)).numpy()
                for ptag, kw in (("dflt", {}), ("tv", dict(tau=2e-6, mu2=1e-4))):
                    rec = ADMM(t(psf).to(tdt), dtype=dtype, norm=norm, **kw)
                    rec.set_data(t(data).to(tdt))
                    rec.reset()
                    s0, s1 = [int(v) for v in rec._convolver._start_idx]
                    for i in range(20):
                        rec._update(i)
                        if i + 1 in (6, 20) and (ptag == "dflt" or i + 1 == 20):
                            r[f"admm_{ptag}_it{i + 1}"] = rec._image_est.numpy()[0, :, s0:s0 + h, s1:s1 + w].copy()
                            if i + 1 == 6:
                                r[f"admm_{ptag}_it6_HV"] = rec._forward_out.numpy()[0, :, s0:s0 + h, s1:s1 + w].copy()
                fis = FISTA(t(psf).to(tdt), dtype=dtype, norm=norm)
                fis.set_data(t(data).to(tdt))
                fis.reset()
                for i in range(40):
                    fis._update(i)
                r["fista_it40"] = fis._image_est.numpy()[0].copy()
            for k, v in res["f64"].items():
                out[f"s{si}_{tag}_{k}"] = v
                out[f"s{si}_{tag}_{k}_ref32"] = np.array(dist(res["f32"][k], v))
            print(name, si, tag, {k: f"{float(out[f's{si}_{tag}_{k}_ref32']):.1e}" for k in res["f64"]})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def operator_case():
    rng = np.random.default_rng(5)
    out = {}
    # next_fast_len table (fftpack flavour, rfft_convolve.py:112)
    ns = np.arange(1, 700)
    out["nfl_n"] = ns
    out["nfl_m"] = np.array([next_fast_len(int(n)) for n in ns])
    out["nfl_big_n"] = np.array([6079, 8111, 539, 959, 2159, 3839, 13, 27, 93, 57])
    out["nfl_big_m"] = np.array([next_fast_len(int(n)) for n in out["nfl_big_n"]])
    for tag, (d, h, w, c) in {"a": (1, 24, 32, 3), "b": (5, 47, 29, 3), "c": (1, 8, 14, 1)}.items():
        psf = rng.random((d, h, w, c)).astype(np.float32)
        x = rng.random((2, d, h, w, c)).astype(np.float32)
        for norm in ("ortho", "backward"):
            cv = RealFFTConvolve2D(t(psf), pad=True, norm=norm)
            out[f"{tag}_{norm}_H"] = cv._H.numpy()
            out[f"{tag}_{norm}_conv"] = cv.convolve(t(x)).numpy()
            out[f"{tag}_{norm}_deconv"] = cv.deconvolve(t(x)).numpy()
            out[f"{tag}_{norm}_pad"] = cv._pad(t(x)).numpy()
        cvn = RealFFTConvolve2D(t(psf), pad=False, norm="backward")
        xp = rng.random([2] + [int(v) for v in cvn._padded_shape]).astype(np.float32)
        out[f"{tag}_xp"] = xp
        out[f"{tag}_nopad_conv"] = cvn.convolve(t(xp)).numpy()
        out[f"{tag}_nopad_deconv"] = cvn.deconvolve(t(xp)).numpy()
        out[f"{tag}_psf"] = psf
        out[f"{tag}_x"] = x
        out[f"{tag}_padded_shape"] = np.array([int(v) for v in cv._padded_shape])
        out[f"{tag}_start"] = np.array([int(v) for v in cv._start_idx])
    v = rng.standard_normal((1, 1, 12, 10, 3)).astype(np.float32)
    u = rng.standard_normal((1, 1, 12, 10, 3, 2)).astype(np.float32)
    out["fd_in"] = v
    out["fd_out"] = finite_diff(t(v)).numpy()
    out["fda_in"] = u
    out["fda_out"] = finite_diff_adj(t(u)).numpy()
    out["st_out"] = soft_thresh(t(u), 0.3).numpy()
    out["gram_12_10_3"] = finite_diff_gram([1, 12, 10, 3], torch.float32, True).numpy()
    out["gram_15_27_1"] = finite_diff_gram([1, 15, 27, 1], torch.float32, True).numpy()
    np.savez_compressed(os.path.join(OUT, "operators.npz"), **out)
    print("wrote operators")


def return_fft_case(name):
    """convolve / deconvolve(return_fft=True): `rfft2(pad(x)) * H` resp. `* conj(H)`, the spectrum the reference hands
    back before its irfft2 (rfft_convolve.py:148-150,161-163,193-195,206-208)."""
    rng = np.random.default_rng(77)
    out = {}
    for tag, (d, h, w, c) in {"a": (2, 12, 20, 3), "b": (1, 9, 13, 1)}.items():
        psf = rng.random((d, h, w, c)).astype(np.float32)
        x = rng.standard_normal((2, d, h, w, c)).astype(np.float32)
        out[f"{tag}_psf"], out[f"{tag}_x"] = psf, x
        for norm in ("ortho", "backward"):
            cv = RealFFTConvolve2D(t(psf), pad=True, norm=norm)
            out[f"{tag}_{norm}_conv_fft"] = cv.convolve(t(x), return_fft=True).numpy()
            out[f"{tag}_{norm}_deconv_fft"] = cv.deconvolve(t(x), return_fft=True).numpy()
        cvn = RealFFTConvolve2D(t(psf), pad=False, norm="backward")
        xp = rng.standard_normal([2] + [int(v) for v in cvn._padded_shape]).astype(np.float32)
        out[f"{tag}_xp"] = xp
        out[f"{tag}_nopad_conv_fft"] = cvn.convolve(t(xp), return_fft=True).numpy()
        out[f"{tag}_nopad_deconv_fft"] = cvn.deconvolve(t(xp), return_fft=True).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if k.endswith("_fft")})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "norm_scale":
        norm_scale_case("norm_scale_ladder")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "return_fft":
        return_fft_case("return_fft")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "unrolled":
        unrolled_admm_case("unrolled_admm_24x32x3_b3", 24, 32, 3, seed=21, n_iter=6, batch=3)
        unrolled_fista_case("unrolled_fista_24x32x3_b3", 24, 32, 3, seed=22, n_iter=7, batch=3)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "admm_pnp":
        admm_pnp_case("pnp_admm")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "hook":
        hook_case("pnp_hook")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "preprocess":
        preprocess_case("preprocess")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "custom_psi":
        custom_psi_case("custom_psi_admm")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "display":
        display_case("apply_display")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "caller_flow":
        caller_flow_case("caller_flow")
        gd_case("fista_profile_gray", FISTA, 38, 50, 1, seed=18, iters=[300])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "recon_error":
        recon_error_case("recon_error")
        sys.exit(0)
    operator_case()
    # default hyper-parameters (U stays 0: tau/mu2 = 10, SURVEY section 7 caveat)
    admm_case("admm_24x32x3_default", 24, 32, 3, seed=1, iters=[1, 2, 5, 20])
    # parameters that exercise the non-zero soft-threshold branch
    admm_case("admm_24x32x3_tv", 24, 32, 3, seed=2, iters=[1, 2, 5, 20, 50], tau=2e-6, mu2=1e-4,
              two_stage=(7, 13))
    # the reference's own test dims (test/test_convolver.py:12), non-square, odd
    admm_case("admm_47x29x3_tv", 47, 29, 3, seed=3, iters=[1, 5, 20], tau=1e-6, mu2=5e-5)
    # odd padded sizes 15 x 27, grayscale
    admm_case("admm_8x14x1_tv", 8, 14, 1, seed=4, iters=[1, 5, 20], tau=1e-6, mu2=5e-5)
    # warm start + background subtraction
    admm_case("admm_24x32x3_init_bg", 24, 32, 3, seed=5, iters=[1, 5, 10], tau=2e-6, mu2=1e-4,
              initial_est=True, background=True)
    admm_case("admm_24x32x1_f64", 24, 32, 1, seed=6, iters=[5, 20], dtype="float64", tau=2e-6, mu2=1e-4)
    # profile/admm.py settings (n_iter=5, gray, float32) at reduced size
    admm_case("admm_profile_gray", 38, 50, 1, seed=7, iters=[5])

    for nm, cls in (("gd", GradientDescent), ("nesterov", NesterovGradientDescent), ("fista", FISTA)):
        gd_case(f"{nm}_24x32x3", cls, 24, 32, 3, seed=11, iters=[1, 2, 5, 20, 60])
        gd_case(f"{nm}_8x14x1", cls, 8, 14, 1, seed=12, iters=[1, 5, 20])
    gd_case("fista_47x29x3_d3", FISTA, 47, 29, 3, seed=13, iters=[1, 5, 20], d=3)
    gd_case("fista_24x32x3_init", FISTA, 24, 32, 3, seed=14, iters=[1, 5, 20], initial_est=True)
    gd_case("fista_24x32x3_tk", FISTA, 24, 32, 3, seed=15, iters=[5, 20], tk=2.5)
    gd_case("nesterov_24x32x3_mu", NesterovGradientDescent, 24, 32, 3, seed=16, iters=[5, 20], mu=0.7)
    gd_case("fista_24x32x1_f64", FISTA, 24, 32, 1, seed=17, iters=[5, 20], dtype="float64")
    unrolled_admm_case("unrolled_admm_24x32x3_b3", 24, 32, 3, seed=21, n_iter=6, batch=3)
    unrolled_fista_case("unrolled_fista_24x32x3_b3", 24, 32, 3, seed=22, n_iter=7, batch=3)
    recon_error_case("recon_error")
    preprocess_case("preprocess")
    hook_case("pnp_hook")
    admm_pnp_case("pnp_admm")
    display_case("apply_display")
    custom_psi_case("custom_psi_admm")
    caller_flow_case("caller_flow")
    norm_scale_case("norm_scale_ladder")
    return_fft_case("return_fft")
    # profile/gradient_descent.py settings (n_iter=300, gray, float32) at reduced size
    gd_case("fista_profile_gray", FISTA, 38, 50, 1, seed=18, iters=[300])
