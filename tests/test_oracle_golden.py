"""
Pins the CPU oracle (oracle/lensless_oracle.py) to golden vectors that were
produced by the real reference (tests/golden/gen_golden.py, build container only).

Tolerances are relative to max|reference|.  The oracle
runs the same torch-CPU FFT library as the reference did when the vectors were
made, so agreement is typically exact; the stated bounds are the contract.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import lensless_oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    if isinstance(a, torch.Tensor):
        a = a.numpy()
    a, b = np.asarray(a), np.asarray(b)
    if np.iscomplexobj(a) or np.iscomplexobj(b):
        a, b = a.astype(np.complex128), b.astype(np.complex128)
    else:
        a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_next_fast_len_table():
    g = np.load(os.path.join(GOLDEN, "operators.npz"))
    for n, m in zip(g["nfl_n"], g["nfl_m"]):
        assert orc.next_fast_len_5smooth(int(n)) == int(m)
    for n, m in zip(g["nfl_big_n"], g["nfl_big_m"]):
        assert orc.next_fast_len_5smooth(int(n)) == int(m)
    # the BASELINE.json sizes
    assert orc.Geometry(3040, 4056).hp == 6144 and orc.Geometry(3040, 4056).wp == 8192
    assert orc.Geometry(270, 480).hp == 540 and orc.Geometry(270, 480).wp == 960
    assert orc.Geometry(1080, 1920).hp == 2160 and orc.Geometry(1080, 1920).wp == 3840


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_operator_vectors(tag):
    g = np.load(os.path.join(GOLDEN, "operators.npz"))
    psf, x = g[f"{tag}_psf"], torch.from_numpy(g[f"{tag}_x"])
    for norm in ("ortho", "backward"):
        cv = orc.ConvolverOracle(psf, pad=True, norm=norm)
        assert [int(v) for v in g[f"{tag}_padded_shape"][1:3]] == [cv.geom.hp, cv.geom.wp]
        assert list(g[f"{tag}_start"]) == [cv.geom.sh, cv.geom.sw]
        assert rel(cv.H, g[f"{tag}_{norm}_H"]) <= 2e-6
        assert rel(cv.convolve(x), g[f"{tag}_{norm}_conv"]) <= 2e-6
        assert rel(cv.deconvolve(x), g[f"{tag}_{norm}_deconv"]) <= 2e-6
        assert np.array_equal(cv.geom.pad(x).numpy(), g[f"{tag}_{norm}_pad"])
        assert torch.equal(cv.geom.crop(cv.geom.pad(x)), x)  # test/test_convolver.py:11-29
    cvn = orc.ConvolverOracle(psf, pad=False, norm="backward")
    xp = torch.from_numpy(g[f"{tag}_xp"])
    assert rel(cvn.convolve(xp), g[f"{tag}_nopad_conv"]) <= 2e-6
    assert rel(cvn.deconvolve(xp), g[f"{tag}_nopad_deconv"]) <= 2e-6


def test_tv_helpers():
    g = np.load(os.path.join(GOLDEN, "operators.npz"))
    assert np.array_equal(orc.finite_diff(torch.from_numpy(g["fd_in"])).numpy(), g["fd_out"])
    assert np.array_equal(orc.finite_diff_adj(torch.from_numpy(g["fda_in"])).numpy(), g["fda_out"])
    assert np.array_equal(orc.soft_thresh(torch.from_numpy(g["fda_in"]), 0.3).numpy(), g["st_out"])
    for key, shp in (("gram_12_10_3", [1, 12, 10, 3]), ("gram_15_27_1", [1, 15, 27, 1])):
        got = orc.finite_diff_gram(shp, torch.float32).numpy()
        assert rel(got, g[key]) <= 2e-6


ADMM_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "admm_*.npz")))


@pytest.mark.parametrize("name", ADMM_CASES)
def test_admm_trajectory(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    dtype = torch.float32 if str(g["dtype"]) == "float32" else torch.float64
    mu1, mu2, mu3, tau = [float(v) for v in g["params"]]
    init = g["initial_est"].copy() if "initial_est" in g else None
    o = orc.ADMMOracle(g["psf"], dtype=dtype, mu1=mu1, mu2=mu2, mu3=mu3, tau=tau, initial_est=init)
    o.set_data(g["data"])
    o.reset()
    if "background" in g:
        o.data = o.data - torch.from_numpy(g["background"]).to(dtype)
        o.data[o.data < 0] = 0
    iters = [int(i) for i in g["iters"]]
    tol = {1: 1e-6, 2: 1e-6, 5: 1e-6, 10: 5e-6, 20: 1e-5, 50: 5e-5}
    for i in range(max(iters)):
        o.step()
        if (i + 1) in iters:
            for key, val in (("V", o.V), ("X", o.X), ("U", o.U), ("W", o.W), ("xi", o.xi),
                             ("eta", o.eta), ("rho", o.rho), ("HV", o.HV)):
                ref = g[f"it{i + 1}_{key}"]
                if np.max(np.abs(ref)) == 0:
                    assert float(val.abs().max()) == 0.0, (key, i + 1)
                else:
                    assert rel(val, ref) <= tol[i + 1], (key, i + 1, rel(val, ref))
    assert rel(o.form_image()[0], g["final"]) <= tol[max(iters)]
    if "two_stage" in g:
        n1, n2 = [int(v) for v in g["two_stage"]]
        o2 = orc.ADMMOracle(g["psf"], dtype=dtype, mu1=mu1, mu2=mu2, mu3=mu3, tau=tau)
        o2.set_data(g["data"])
        o2.apply(n1)
        res = o2.apply(n2, reset=False)
        assert rel(res, g["two_stage_final"]) <= 1e-5


def test_admm_tv_case_exercises_nonzero_U():
    g = np.load(os.path.join(GOLDEN, "admm_24x32x3_tv.npz"))
    assert np.count_nonzero(g["it20_U"]) > 0.05 * g["it20_U"].size
    g0 = np.load(os.path.join(GOLDEN, "admm_24x32x3_default.npz"))
    assert np.count_nonzero(g0["it20_U"]) == 0  # SURVEY section 7 caveat


GD_CASES = sorted(
    os.path.basename(p)[:-4]
    for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
    if os.path.basename(p).split("_")[0] in ("gd", "nesterov", "fista")
)


@pytest.mark.parametrize("name", GD_CASES)
def test_gd_family_trajectory(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kind = {"gd": "vanilla", "nesterov": "nesterov", "fista": "fista"}[name.split("_")[0]]
    dtype = torch.float32 if str(g["dtype"]) == "float32" else torch.float64
    kw = {}
    if name.endswith("_tk"):
        kw["tk"] = 2.5
    if name.endswith("_mu"):
        kw["mu"] = 0.7
    init = g["initial_est"].copy() if "initial_est" in g else None
    o = orc.GDOracle(g["psf"], kind=kind, dtype=dtype, initial_est=init, **kw)
    o.set_data(g["data"])
    assert rel(o.alpha, g["alpha"]) <= 2e-6
    assert rel(o.x, g["x0"]) <= 1e-7
    iters = [int(i) for i in g["iters"]]
    for i in range(max(iters)):
        o.step()
        if (i + 1) in iters:
            r = rel(o.x, g[f"it{i + 1}_x"])
            assert r <= (2e-6 if i < 5 else 5e-5), (i + 1, r)
    assert rel(o.form_image()[0], g["final"]) <= 5e-5


def test_unrolled_admm_schedule_matches_reference():
    """UnrolledADMM.forward on a batch of 3 with different parameters in every iteration."""
    g = np.load(os.path.join(GOLDEN, "unrolled_admm_24x32x3_b3.npz"))
    sched = {k: g[k] for k in ("mu1", "mu2", "mu3", "tau")}
    for b in range(g["data"].shape[0]):
        o = orc.ADMMOracle(g["psf"], schedule=sched)
        o.set_data(g["data"][b, 0])
        assert rel(o.apply(int(g["n_iter"])), g["out"][b]) <= 2e-6


def test_unrolled_fista_matches_reference():
    g = np.load(os.path.join(GOLDEN, "unrolled_fista_24x32x3_b3.npz"))
    for b in range(g["data"].shape[0]):
        out = orc.unrolled_fista_oracle(g["psf"], g["data"][b, 0], g["alpha"], g["tk"])
        assert rel(out[0], g["out"][b]) <= 2e-6


def test_reconstruction_error_matches_reference():
    """recon.py:607-653 values produced by the imported reference (gen_golden.py recon_error)."""
    g = np.load(os.path.join(GOLDEN, "recon_error.npz"))
    tau, mu2 = (float(v) for v in g["admm_params"])
    o = orc.ADMMOracle(g["admm_psf"], tau=tau, mu2=mu2)
    o.set_data(g["admm_data"])
    pred = o.apply(int(g["admm_iters"]))[None]
    conv = orc.ConvolverOracle(g["admm_psf"], pad=True, norm="backward")
    assert rel(orc.reconstruction_error(conv, pred, o.data), g["admm_err"]) <= 2e-5
    assert rel(orc.reconstruction_error(conv, pred, o.data, normalize=False), g["admm_err_raw"]) <= 2e-5
    f = orc.GDOracle(g["fista_psf"], kind="fista")
    f.set_data(g["fista_data"])
    pred = f.apply(int(g["fista_iters"]))[None]
    assert rel(orc.reconstruction_error(f.conv, pred, f.data), g["fista_err"]) <= 2e-5
    assert rel(orc.reconstruction_error(f.conv, pred, f.data, normalize=False), g["fista_err_raw"]) <= 2e-5
    conv3 = orc.ConvolverOracle(g["x_psf"], pad=True, norm="ortho")
    err = orc.reconstruction_error(conv3, torch.from_numpy(g["x_pred"]), torch.from_numpy(g["x_frames"]))
    assert rel(err, g["x_err"]) <= 2e-5


def _shrink(x):
    return torch.clamp(x - 0.05, min=0)


def test_custom_projection_and_denoiser_hook_match_reference():
    """gd.py:67,89-92,136-140: `proj=` callables / a denoiser standing in for the projection."""
    g = np.load(os.path.join(GOLDEN, "pnp_hook.npz"))
    n = int(g["iters"])
    for nm, kind in (("gd", "vanilla"), ("nesterov", "nesterov"), ("fista", "fista")):
        o = orc.GDOracle(g["psf"], kind=kind, proj=_shrink)
        o.set_data(g["data"])
        assert rel(o.apply(n), g[nm + "_final"]) <= 1e-6
        assert rel(o.x, g[nm + "_state"]) <= 1e-6
    nl = float(g["noise_level"])
    o = orc.GDOracle(g["psf"], kind="fista", proj=lambda x: torch.clamp(x, min=0) * (1.0 - nl / 100.0))
    o.set_data(g["data"])
    assert rel(o.apply(n), g["pnp_final"]) <= 1e-6


def _pnp_denoise(x, noise_level):
    sm = 0.5 * x + 0.25 * (torch.roll(x, 1, dims=-2) + torch.roll(x, -1, dims=-2))
    return sm * (1.0 - noise_level / 200.0)


@pytest.mark.parametrize("dual", [False, True])
def test_admm_plug_and_play_branch_matches_reference(dual):
    """admm.py:126-133,235-243,266-275,300-311 with a stand-in denoiser, both use_dual settings, including the
    continuation after the in-place clamp of _form_image."""
    g = np.load(os.path.join(GOLDEN, "pnp_admm.npz"))
    mu1, mu2, mu3 = (float(v) for v in g["params"])
    o = orc.ADMMOracle(g["psf"], mu1=mu1, mu2=mu2, mu3=mu3, initial_est=g["initial_est"].copy(),
                       denoiser=(_pnp_denoise, float(g["noise_level"]), dual))
    o.set_data(g["data"])
    tag = "dual" if dual else "plain"
    assert rel(o.apply(int(g["iters"])), g[tag + "_final"]) <= 1e-5
    for k, a in (("_image_est", o.V), ("_U", o.U), ("_X", o.X), ("_W", o.W), ("_xi", o.xi), ("_eta", o.eta),
                 ("_rho", o.rho)):
        assert rel(a, g[tag + k]) <= 1e-5, k
    assert rel(o.apply(3, reset=False), g[tag + "_more"]) <= 1e-5
