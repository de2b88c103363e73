"""
Opt-in long anchors at 12 MP (`pytest -m gpu_long`; about 10 minutes of host time for the float64 oracle).  The driver's
`-m gpu` step does not select them -- that step has 20 minutes for 250 tests -- and the CPU suite skips them (no device).
Their last run on the MI355X is recorded in profiles/r05_gpu_long.log.

What they close: tests/test_parity_fullsize.py asserts the 100-iteration ADMM call, the 300-iteration FISTA call and the
50-iteration depth stack against the FLOAT64 BUILD of the engine, which itself is compared with the float64 oracle for
5 / 6 / 12 iterations.  Same sources in both builds: a length-dependent logic error (the steady-state path of
lpc_iterate(), the half-applied duals, the sensor-window structure) would cancel in that comparison.  Here the float64
build runs 30 ADMM iterations in ONE call -- 26 of them on the steady-state path -- and 40 FISTA iterations against the
float64 ORACLE itself (reference loop: /root/reference/lensless/recon/recon.py:575-576 over admm.py:313-338 and
gd.py:235-241).
"""
import os

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc

pytestmark = [pytest.mark.gpu_long, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a MI355X")]


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def inputs():
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    H, W, C = 3040, 4056, 3
    psf = orc.synthetic_psf(1, H, W, C, seed=0)
    scene = orc.synthetic_scene(H, W, C, seed=1)
    cv = lpa.RealFFTConvolve2D(torch.from_numpy(psf).cuda(), pad=True, norm="backward")
    y = cv.convolve(torch.from_numpy(scene).cuda()[None, None])[0, 0].clamp_(min=0)
    y = (y / y.max()).contiguous().cpu().numpy()
    del cv
    torch.cuda.empty_cache()
    return psf, y


@pytest.mark.parametrize("plain", [False, True], ids=["window_structure", "plain"])
def test_float64_build_vs_float64_oracle_30_admm_iterations(inputs, plain):
    psf, y = inputs
    kw = dict(tau=2e-9, mu2=1e-4)            # TV-active at this size (tests/test_parity_fullsize.py walks the ladder)
    rec = lpa.ADMM(torch.from_numpy(psf).cuda().double(), dtype="float64",
                   engine_options={"hv_full": 1, "xi_full": 1} if plain else {}, **kw)
    rec.set_data(torch.from_numpy(y).cuda().double())
    got = rec.apply(n_iter=30, disp_iter=None).cpu().numpy()
    info = rec._handle.plan_info()
    del rec
    torch.cuda.empty_cache()
    assert ("row transforms skipped" in info) != plain, info
    o = orc.ADMMOracle(psf, dtype=torch.float64, **kw)
    o.set_data(y)
    ref = o.apply(30).numpy()
    nz = float((o.U != 0).double().mean())
    e = rel(got, ref)
    print(f"float64 build ({'plain' if plain else 'steady-state path'}) vs float64 oracle, ADMM 30 iterations at 12 MP: {e:.2e}; "
          f"U non-zero {100 * nz:.1f} %")
    assert 0.02 < nz < 0.999 and e <= 1e-9, (nz, e)


def test_float64_build_vs_float64_oracle_40_fista_iterations(inputs):
    psf, y = inputs
    rec = lpa.FISTA(torch.from_numpy(psf).cuda().double(), dtype="float64")
    rec.set_data(torch.from_numpy(y).cuda().double())
    got = rec.apply(n_iter=40, disp_iter=None).cpu().numpy()
    del rec
    torch.cuda.empty_cache()
    o = orc.GDOracle(psf, kind="fista", dtype=torch.float64)
    o.set_data(y)
    e = rel(got, o.apply(40).numpy())
    print(f"float64 build vs float64 oracle, FISTA 40 iterations at 12 MP: {e:.2e}")
    assert e <= 1e-9, e
