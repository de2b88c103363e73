"""Behavioural tests of the mirrored plugin API beyond plain apply(): warm starts, PSF swaps,
momentum resets, reconstruction_error, the array-level convenience wrappers and the plot/save hooks."""
import os

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture
def small():
    psf = orc.synthetic_psf(1, 20, 28, 3, seed=2)
    y = np.random.default_rng(2).random((20, 28, 3), dtype=np.float32)
    return psf, y


def test_set_psf_swaps_operator_and_resets(backend, small):
    psf, y = small
    psf2 = orc.synthetic_psf(1, 20, 28, 3, seed=9)
    rec = lpa.FISTA(psf)
    rec.set_data(y)
    rec.apply(n_iter=3, disp_iter=None)
    rec._set_psf(psf2)                                        # recon.py:448-470 (used by multimask benchmarks)
    got = rec.apply(n_iter=4, disp_iter=None)
    o = orc.GDOracle(psf2, kind="fista")
    o.set_data(y)
    assert rel(got, o.apply(4)) <= 5e-6
    with pytest.raises(AssertionError):
        rec._set_psf(psf2[:, :10])


def test_warm_start_via_set_image_estimate(backend, small):
    psf, y = small
    est = np.random.default_rng(5).random((1, 1, 20, 28, 3), dtype=np.float32)
    rec = lpa.GradientDescent(psf)
    rec.set_data(y)
    rec.set_image_estimate(est)
    assert rel(rec._image_est, est) == 0.0
    got = rec.apply(n_iter=5, disp_iter=None)                 # apply() resets -> restarts from the estimate
    o = orc.GDOracle(psf, kind="vanilla", initial_est=est)
    o.set_data(y)
    assert rel(got, o.apply(5)) <= 5e-6


def test_nesterov_and_fista_reset_arguments(backend, small):
    psf, y = small
    nes = lpa.NesterovGradientDescent(psf)
    nes.set_data(y)
    nes.reset(p=0, mu=0.5)                                    # gd.py:178-181
    nes.apply(n_iter=6, disp_iter=None, reset=False)
    o = orc.GDOracle(psf, kind="nesterov")
    o.set_data(y)
    o.reset(mu=0.5)
    for _ in range(6):
        o.step()
    assert rel(nes._image_est, o.x) <= 5e-6
    fis = lpa.FISTA(psf)
    fis.set_data(y)
    fis.reset(tk=3.0)                                         # gd.py:227-233
    fis.apply(n_iter=6, disp_iter=None, reset=False)
    o2 = orc.GDOracle(psf, kind="fista")
    o2.set_data(y)
    o2.reset(tk=3.0)
    for _ in range(6):
        o2.step()
    assert rel(fis._image_est, o2.x) <= 5e-6


def test_reconstruction_error_matches_oracle(backend, small):
    psf, y = small
    rec = lpa.FISTA(torch.from_numpy(psf))
    rec.set_data(torch.from_numpy(y))
    rec.apply(n_iter=5, disp_iter=None)
    err = rec.reconstruction_error()
    o = orc.GDOracle(psf, kind="fista")
    o.set_data(y)
    pred = o.apply(5)[None]
    ref = orc.reconstruction_error(o.conv, pred, o.data)
    assert isinstance(err, torch.Tensor) and err.shape == (1,)
    assert rel(err, ref) <= 1e-4


def test_array_level_wrappers(backend, small):
    psf, y = small
    res = lpa.apply_admm(psf, y, n_iter=4, tau=2e-6, mu2=1e-4)
    o = orc.ADMMOracle(psf, tau=2e-6, mu2=1e-4)
    o.set_data(y)
    assert rel(res, o.apply(4)) <= 5e-6
    res2 = lpa.apply_gradient_descent(psf, y, n_iter=4)
    o2 = orc.GDOracle(psf, kind="vanilla")
    o2.set_data(y)
    assert rel(res2, o2.apply(4)) <= 5e-6


def test_plot_and_save_hooks(backend, small, tmp_path):
    pytest.importorskip("matplotlib")
    psf, y = small
    rec = lpa.ADMM(psf)
    rec.set_data(y)
    out = rec.apply(n_iter=6, disp_iter=3, plot=True, save=str(tmp_path), plot_pause=0.0)
    assert isinstance(out, tuple) and out[0].shape == (1, 20, 28, 3)      # (image, ax) like recon.py:594-604
    assert sorted(os.listdir(tmp_path)) == ["3.png", "6.png"]
    plain = lpa.ADMM(psf)
    plain.set_data(y)
    # displaying after iteration 3 clamps the state in place, in the reference (recon.py:580-583 ->
    # admm.py:331-338) and here alike: the result equals apply(3) + apply(3, reset=False), not apply(6)
    staged = lpa.ADMM(psf)
    staged.set_data(y)
    staged.apply(n_iter=3, disp_iter=None)
    assert np.array_equal(staged.apply(n_iter=3, disp_iter=None, reset=False), out[0])


def test_background_subtraction_is_cumulative_like_reference(backend, small):
    psf, y = small
    bg = np.full_like(y, 0.2)
    rec = lpa.GradientDescent(psf)
    rec.set_data(y)
    a = rec.apply(n_iter=3, disp_iter=None, background=bg)
    b = rec.apply(n_iter=3, disp_iter=None, background=bg)    # recon.py:553-555 mutates self._data again
    o = orc.GDOracle(psf, kind="vanilla")
    o.set_data(y)
    ra = o.apply(3, background=bg)
    rb = o.apply(3, background=bg)
    assert rel(a, ra) <= 5e-6 and rel(b, rb) <= 5e-6
    assert not np.array_equal(a, b)
