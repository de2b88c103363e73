"""Plug-and-play hook of the gradient-descent family: custom `proj=` callables and
an external denoiser run through lpc_iterate_begin / lpc_iterate_end; checked against the reference's own outputs
(tests/golden/pnp_hook.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from lenslesspicam_amd._native import NativeError
from oracle import lensless_oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))   # all-zero reference: 0 iff equal


def shrink(x):            # same function the golden generator handed to the reference; not idempotent
    return torch.clamp(x - 0.05, min=0)


@pytest.mark.parametrize("nm,cls", [("gd", "GradientDescent"), ("nesterov", "NesterovGradientDescent"),
                                    ("fista", "FISTA")])
def test_custom_projection_golden(backend, nm, cls):
    g = np.load(os.path.join(GOLDEN, "pnp_hook.npz"))
    rec = getattr(lpa, cls)(torch.from_numpy(g["psf"]), proj=shrink)
    rec.set_data(torch.from_numpy(g["data"]))
    got = rec.apply(n_iter=int(g["iters"]), disp_iter=None)
    assert rel(got, g[nm + "_final"]) <= 5e-6
    assert rel(rec._image_est, g[nm + "_state"]) <= 5e-6


def test_denoiser_replaces_projection_golden(backend):
    g = np.load(os.path.join(GOLDEN, "pnp_hook.npz"))
    seen = []

    def denoise(x, noise_level):
        seen.append((tuple(x.shape), float(noise_level)))
        return torch.clamp(x, min=0) * (1.0 - noise_level / 100.0)

    rec = lpa.FISTA(torch.from_numpy(g["psf"]), denoiser={"network": denoise, "noise_level": float(g["noise_level"])})
    rec.set_data(torch.from_numpy(g["data"]))
    got = rec.apply(n_iter=int(g["iters"]), disp_iter=None)
    assert rel(got, g["pnp_final"]) <= 5e-6
    assert len(seen) == int(g["iters"]) + 1 and seen[0] == ((1, 1) + g["psf"].shape[1:], 7.0)
    with pytest.raises(NotImplementedError):
        lpa.FISTA(torch.from_numpy(g["psf"]), denoiser={"network": "DruNet", "noise_level": 10})


def test_numpy_callable_float64_tk_and_continue(backend):
    psf = orc.synthetic_psf(2, 14, 18, 1, seed=3).astype(np.float64)
    y = np.random.default_rng(3).random((14, 18, 1))

    def box(x):           # NumPy in, NumPy out when the solver was built from NumPy arrays
        assert isinstance(x, np.ndarray) and x.dtype == np.float64
        return np.clip(x, 0.0, 0.4)

    rec = lpa.FISTA(psf, dtype="float64", proj=box, tk=2.0)
    rec.set_data(y)
    rec.apply(n_iter=3, disp_iter=None)
    got = rec.apply(n_iter=4, disp_iter=None, reset=False)
    o = orc.GDOracle(psf, kind="fista", dtype=torch.float64, tk=2.0, proj=lambda x: torch.clamp(x, 0.0, 0.4))
    o.set_data(y)
    o.apply(3)
    assert rel(got, o.apply(4, reset=False)) <= 1e-12


def test_split_protocol_errors(backend):
    psf = orc.synthetic_psf(1, 12, 16, 1, seed=5)
    rec = lpa.GradientDescent(psf)
    rec.set_data(np.random.default_rng(5).random((12, 16, 1), dtype=np.float32))
    h = rec._handle
    with pytest.raises(NativeError):
        h.iterate_end(rec._psf_dev.data_ptr())          # nothing in flight
    h.iterate_begin()
    with pytest.raises(NativeError):
        h.iterate(1)                                    # fused loop while a split iteration is open
    with pytest.raises(NativeError):
        h.iterate_begin()
    x = rec._empty(rec._state_shape())
    h.get_state("image_est", x.data_ptr())
    proj = torch.clamp(x, min=0).contiguous()            # keep it alive across the call
    h.iterate_end(proj.data_ptr())
    # begin + non_neg + end == one fused iteration
    ref = lpa.GradientDescent(psf)
    ref.set_data(rec._data)
    ref._iterate(1)
    assert rel(rec._image_est, ref._image_est) == 0.0
    adm = lpa.ADMM(psf)
    with pytest.raises(NativeError):
        adm._handle.iterate_begin()


def pnp_denoise(x, noise_level):          # the function the golden generator handed to the reference
    sm = 0.5 * x + 0.25 * (torch.roll(x, 1, dims=-2) + torch.roll(x, -1, dims=-2))
    return sm * (1.0 - noise_level / 200.0)


@pytest.mark.parametrize("dual", [False, True])
def test_admm_plug_and_play_golden(backend, dual):
    """ADMM with an external denoiser as the U-prox (admm.py:126-133,235-243,266-275,300-311) through
    lpc_admm_pnp_begin / lpc_admm_pnp_end, against the reference's own run: final image, every state array, and
    the continuation after the in-place clamp of _form_image."""
    g = np.load(os.path.join(GOLDEN, "pnp_admm.npz"))
    mu1, mu2, mu3 = (float(v) for v in g["params"])
    rec = lpa.ADMM(torch.from_numpy(g["psf"]), mu1=mu1, mu2=mu2, mu3=mu3,
                   initial_est=torch.from_numpy(g["initial_est"].copy()),
                   denoiser={"network": pnp_denoise, "noise_level": float(g["noise_level"]), "use_dual": dual})
    rec.set_data(torch.from_numpy(g["data"]))
    tag = "dual" if dual else "plain"
    tol = 2e-5
    assert rel(rec.apply(n_iter=int(g["iters"]), disp_iter=None), g[tag + "_final"]) <= tol
    for k in ("_image_est", "_U", "_X", "_W", "_xi", "_eta", "_rho"):
        got = getattr(rec, k)
        assert tuple(got.shape) == g[tag + k].shape, k
        assert rel(got, g[tag + k]) <= tol, k
    assert rel(rec.apply(n_iter=3, disp_iter=None, reset=False), g[tag + "_more"]) <= tol


def test_admm_plug_and_play_protocol(backend):
    psf = orc.synthetic_psf(1, 12, 16, 1, seed=6)
    y = np.random.default_rng(6).random((12, 16, 1), dtype=np.float32)
    rec = lpa.ADMM(psf, denoiser={"network": lambda x, s: np.maximum(x, 0) * 0.9, "noise_level": 5, "use_dual": False})
    rec.set_data(y)
    got = rec.apply(n_iter=4, disp_iter=None)                        # NumPy in, NumPy callable, NumPy out
    assert isinstance(got, np.ndarray)
    o = orc.ADMMOracle(psf, denoiser=(lambda x, s: torch.clamp(x, min=0) * 0.9, 5, False))
    o.set_data(y)
    assert rel(got, o.apply(4)) <= 1e-5
    h = rec._handle
    with pytest.raises(NativeError):
        h.iterate(1)                                                 # fused iterations on a plug-and-play handle
    with pytest.raises(NativeError):
        h.admm_pnp_end(False, rec._psf_dev.data_ptr())               # nothing in flight
    plain = lpa.ADMM(psf)
    plain.set_data(y)
    plain._iterate(2)
    with pytest.raises(NativeError):
        x = plain._empty(plain._state_shape())
        plain._handle.admm_pnp_begin(False, x.data_ptr())            # fused iterations already ran
    with pytest.raises(NotImplementedError):
        lpa.ADMM(psf, denoiser={"network": "DruNet", "noise_level": 10, "use_dual": True})
