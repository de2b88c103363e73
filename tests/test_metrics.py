"""Evaluation reductions: reconstruction_error and mse / psnr computed on the
device by lpc_reconstruction_error / lpc_image_metrics, against the reference's own values
(tests/golden/recon_error.npz) and against the oracle."""
import os

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from lenslesspicam_amd import metric
from oracle import lensless_oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5   # float32: one convolution + a sum of ~1e3..1e7 squares, relative


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / max(np.max(np.abs(b)), 1e-300))


def test_reconstruction_error_golden_admm(backend):
    g = np.load(os.path.join(GOLDEN, "recon_error.npz"))
    tau, mu2 = (float(v) for v in g["admm_params"])
    rec = lpa.ADMM(torch.from_numpy(g["admm_psf"]), tau=tau, mu2=mu2)
    rec.set_data(torch.from_numpy(g["admm_data"]))
    rec.apply(n_iter=int(g["admm_iters"]), disp_iter=None)
    err = rec.reconstruction_error()
    assert isinstance(err, torch.Tensor) and err.shape == (1,)
    assert rel(err, g["admm_err"]) <= TOL
    assert rel(rec.reconstruction_error(normalize=False), g["admm_err_raw"]) <= TOL
    # evaluating must not disturb the solver: continue and compare with an uninterrupted run
    more = rec.apply(n_iter=3, disp_iter=None, reset=False)
    o = orc.ADMMOracle(g["admm_psf"], tau=tau, mu2=mu2)
    o.set_data(g["admm_data"])
    o.apply(int(g["admm_iters"]))
    o.form_image()          # the extra get_image_estimate() inside reconstruction_error clamps in place
    assert rel(more, o.apply(3, reset=False)) <= 5e-6


def test_reconstruction_error_golden_fista_depth2(backend):
    g = np.load(os.path.join(GOLDEN, "recon_error.npz"))
    rec = lpa.FISTA(g["fista_psf"])                      # NumPy in -> NumPy out
    rec.set_data(g["fista_data"])
    rec.apply(n_iter=int(g["fista_iters"]), disp_iter=None)
    err = rec.reconstruction_error()
    assert isinstance(err, np.ndarray) and err.shape == (1,)
    assert rel(err, g["fista_err"]) <= TOL
    assert rel(rec.reconstruction_error(normalize=False), g["fista_err_raw"]) <= TOL
    more = rec.apply(n_iter=2, disp_iter=None, reset=False)
    o = orc.GDOracle(g["fista_psf"], kind="fista")
    o.set_data(g["fista_data"])
    o.apply(int(g["fista_iters"]))
    assert rel(more, o.apply(2, reset=False)) <= 5e-6
    # explicit (prediction, lensless, psfs): batch of 2, another PSF (recon.py:622-633)
    err = rec.reconstruction_error(prediction=g["x_pred"], lensless=g["x_frames"], psfs=g["x_psf"])
    assert err.shape == (2,) and rel(err, g["x_err"]) <= TOL


def test_reconstruction_error_batched_and_float64(backend):
    psf = orc.synthetic_psf(1, 18, 22, 1, seed=4)
    ys = np.random.default_rng(4).random((3, 18, 22, 1))
    rec = lpa.FISTA(psf.astype(np.float64), dtype="float64")
    rec.set_data(ys[:, None])
    res = rec.apply_batch(n_iter=4)
    err = rec.reconstruction_error()                     # current estimate against the stored frames
    o = orc.GDOracle(psf, kind="fista", dtype=torch.float64)
    for b in range(3):
        o.set_data(ys[b])
        pred = o.apply(4)[None]
        assert rel(res[b], pred[0]) <= 1e-12
        assert rel(err[b:b + 1], orc.reconstruction_error(o.conv, pred, o.data)) <= 1e-10


@pytest.mark.parametrize("shape", [(1, 33, 47, 3), (3, 16, 20, 1), (2, 300, 401, 3)])
def test_mse_psnr_match_restated_formula(backend, shape):
    rng = np.random.default_rng(7)
    t = rng.random(shape).astype(np.float32) * 3.0
    e = (t + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    out = metric.metrics_batch(torch.from_numpy(t), torch.from_numpy(e)).cpu().numpy()
    for i in range(shape[0]):
        assert abs(out[i, 0] - orc.mse(t[i], e[i])) <= 1e-5 * orc.mse(t[i], e[i])
        assert abs(out[i, 1] - orc.psnr_skimage(t[i], e[i])) <= 1e-4
    raw = metric.metrics_batch(t, e, normalize=False).cpu().numpy()
    assert abs(raw[0, 0] - orc.mse(t[0], e[0], normalize=False)) <= 1e-5 * raw[0, 0]
    assert abs(metric.mse(t[0], e[0]) - orc.mse(t[0], e[0])) <= 1e-5 * orc.mse(t[0], e[0])
    assert abs(metric.psnr(t[0], e[0]) - orc.psnr_skimage(t[0], e[0])) <= 1e-4


def test_psnr_data_range_rule_for_signed_images(backend):
    rng = np.random.default_rng(8)
    t = rng.standard_normal((1, 12, 14, 3)).astype(np.float32)       # min < 0 -> data range 2
    e = (t + 0.1 * rng.standard_normal(t.shape)).astype(np.float32)
    assert abs(metric.psnr(t[0], e[0], normalize=False) - orc.psnr_skimage(t[0], e[0], normalize=False)) <= 1e-4
