"""
GPU-only parity at sizes the oracle still finishes in seconds, plus size-independent properties at
BASELINE.json's full sizes (the oracle needs ~17 s per 12-MP iteration, so full-size checks are
properties, not trajectories):

  * C1 (270x480x3): ADMM 5 / 100 iterations and FISTA 300 iterations vs the oracle, PSNR delta <= 0.01 dB
  * 760x1014 gray (profile/admm.py's frame, inferred): exercises the four-step column split
  * C2 (3040x4056x3): linearity and adjointness of the operator, delta-PSF identity,
    batch/plane independence, zero-data fixed point, exact iteration accounting
  * C5 (16 planes of 1080x1920x3): plane d of the batched ADMM == a D=1 run with psf[d]
"""
import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max())


def synth(H, W, C, seed=0, D=1):
    psf = orc.synthetic_psf(D, H, W, C, seed=seed)
    scene = orc.synthetic_scene(H, W, C, seed=seed + 1)
    y = orc.synthetic_measurement(psf, scene)
    return psf, scene, y


@pytest.fixture(scope="module")
def c1():
    torch.set_num_threads(16)
    return synth(270, 480, 3)


def test_c1_admm_5_and_100_iterations(c1):
    psf, scene, y = c1
    rec = lpa.ADMM(torch.from_numpy(psf).cuda())
    rec.set_data(torch.from_numpy(y).cuda())
    o = orc.ADMMOracle(psf)
    o.set_data(y)
    o64 = orc.ADMMOracle(psf, dtype=torch.float64)          # truth: the float32 backends of the reference
    o64.set_data(y)                                         # differ from each other by ~6e-6 at this size
    g5 = rec.apply(n_iter=5, disp_iter=None)                # profile/admm.py: n_iter=5
    c5, t5 = o.apply(5), o64.apply(5)
    assert rel(g5, c5) <= 1e-5
    assert rel(g5, t5) <= 5e-6 and rel(g5, t5) <= 2 * rel(c5, t5) + 1e-6
    g100 = rec.apply(n_iter=100, disp_iter=None)
    c100, t100 = o.apply(100), o64.apply(100)
    assert rel(g100, c100) <= 5e-5
    assert rel(g100, t100) <= 2e-5 and rel(g100, t100) <= 2 * rel(c100, t100) + 1e-6
    d = orc.psnr(g100[0].cpu().numpy(), scene) - orc.psnr(c100[0].numpy(), scene)
    assert abs(d) <= 0.01


def test_c1_admm_tv_active_100_iterations(c1):
    psf, scene, y = c1
    kw = dict(tau=2e-6, mu2=1e-4)
    rec = lpa.ADMM(torch.from_numpy(psf).cuda(), **kw)
    rec.set_data(torch.from_numpy(y).cuda())
    o = orc.ADMMOracle(psf, **kw)
    o.set_data(y)
    g = rec.apply(n_iter=100, disp_iter=None)
    c = o.apply(100)
    assert float(o.U.abs().max()) > 0                        # the soft-threshold branch is live
    assert rel(g, c) <= 5e-5
    assert abs(orc.psnr(g[0].cpu().numpy(), scene) - orc.psnr(c[0].numpy(), scene)) <= 0.01


@pytest.mark.parametrize("kind,cls", [("fista", lpa.FISTA), ("nesterov", lpa.NesterovGradientDescent),
                                       ("vanilla", lpa.GradientDescent)])
def test_c1_gd_family_300_iterations(c1, kind, cls):
    psf, scene, y = c1
    rec = cls(torch.from_numpy(psf).cuda())
    rec.set_data(torch.from_numpy(y).cuda())
    g = rec.apply(n_iter=300, disp_iter=None)                # profile/gradient_descent.py: 300 iterations
    o = orc.GDOracle(psf, kind=kind)
    o.set_data(y)
    c = o.apply(300)
    assert rel(g, c) <= 5e-4
    assert abs(orc.psnr(g[0].cpu().numpy(), scene) - orc.psnr(c[0].numpy(), scene)) <= 0.01


def test_split_columns_760x1014_gray():
    """1519 -> 1536 rows: too long for one LDS tile => four-step split path on the GPU."""
    torch.set_num_threads(16)
    psf, scene, y = synth(760, 1014, 1, seed=3)
    rec = lpa.ADMM(torch.from_numpy(psf).cuda(), tau=2e-6, mu2=1e-4)
    assert rec._padded_shape[1:3] == [1536, 2048]
    rec.set_data(torch.from_numpy(y).cuda())
    g = rec.apply(n_iter=5, disp_iter=None)
    # At this size the float32 CPU backend of the reference/oracle is itself ~3e-5 away from float64
    # truth (tools/accuracy_probe.py, profiles/r01a_accuracy.log), so truth = the float64 oracle:
    # the engine must be within 1e-5 of it AND no further from it than the float32 oracle is.
    o64 = orc.ADMMOracle(psf, dtype=torch.float64, tau=2e-6, mu2=1e-4)
    o64.set_data(y)
    t64 = o64.apply(5)
    o32 = orc.ADMMOracle(psf, tau=2e-6, mu2=1e-4)
    o32.set_data(y)
    e_gpu, e_cpu = rel(g, t64), rel(o32.apply(5), t64)
    assert e_gpu <= 1e-5 and e_gpu <= 2 * e_cpu, (e_gpu, e_cpu)
    f = lpa.FISTA(torch.from_numpy(psf).cuda())
    f.set_data(torch.from_numpy(y).cuda())
    of = orc.GDOracle(psf, kind="fista")
    of.set_data(y)
    assert rel(f.apply(n_iter=10, disp_iter=None), of.apply(10)) <= 1e-5
    cv = lpa.RealFFTConvolve2D(torch.from_numpy(psf).cuda(), pad=True)
    oc = orc.ConvolverOracle(psf, pad=True)
    x = torch.from_numpy(scene)[None, None]
    assert rel(cv.convolve(x.cuda()), oc.convolve(x)) <= 2e-6
    assert rel(cv.deconvolve(x.cuda()), oc.deconvolve(x)) <= 2e-6


# ------------------------------------------------------------------ full size: properties --
@pytest.fixture(scope="module")
def c2():
    H, W, C = 3040, 4056, 3
    g = torch.Generator(device="cuda").manual_seed(0)
    psf = torch.rand((1, H, W, C), device="cuda", generator=g) ** 12
    psf /= psf.norm()
    return H, W, C, psf, g


def test_c2_operator_linearity_adjointness_and_delta(c2):
    H, W, C, psf, g = c2
    cv = lpa.RealFFTConvolve2D(psf, pad=True, norm="ortho")
    assert cv._padded_shape == [1, 6144, 8192, 3]
    x1 = torch.randn((1, 1, H, W, C), device="cuda", generator=g)
    x2 = torch.randn((1, 1, H, W, C), device="cuda", generator=g)
    a, b = 0.37, -1.9
    lhs = cv.convolve(a * x1 + b * x2)
    rhs = a * cv.convolve(x1) + b * cv.convolve(x2)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) <= 5e-6
    y = torch.randn((1, 1, H, W, C), device="cuda", generator=g)
    dot1 = float((cv.convolve(x1).double() * y.double()).sum())
    dot2 = float((x1.double() * cv.deconvolve(y).double()).sum())
    assert abs(dot1 - dot2) <= 1e-5 * max(abs(dot1), abs(dot2))
    # a delta PSF at the window centre makes the operator a scaled identity
    dpsf = torch.zeros_like(psf)
    dpsf[0, H // 2, W // 2, :] = 1.0
    cd = lpa.RealFFTConvolve2D(dpsf, pad=True, norm="backward")
    # one impulse in -> one impulse out (a pure translate), nothing else anywhere in the 12-MP frame
    e = torch.zeros_like(x1)
    e[0, 0, 100, 200, :] = 1.0
    r = cd.convolve(e)[0, 0, :, :, 0]
    pos = int(torch.argmax(r))
    assert abs(float(r.flatten()[pos]) - 1.0) <= 1e-5
    r.flatten()[pos] = 0
    assert float(r.abs().max()) <= 1e-5


def test_c2_admm_zero_data_fixed_point_and_accounting(c2):
    H, W, C, psf, g = c2
    rec = lpa.ADMM(psf)
    rec.set_data(torch.zeros((H, W, C), device="cuda"))
    out = rec.apply(n_iter=3, disp_iter=None)
    assert float(out.abs().max()) == 0.0                      # y = 0, V0 = 0 is a fixed point of every update
    assert rec._handle is not None
    y = torch.rand((H, W, C), device="cuda", generator=g)
    rec.set_data(y)
    a = rec.apply(n_iter=4, disp_iter=None).clone()
    rec.reset()
    rec._iterate(2)
    rec._iterate(2)                                           # 2 + 2 launches == 4: exact iteration accounting
    b = rec.get_image_estimate()[0]
    # (not bit-equal: inside ONE call the rows of H V outside the sensor window skip their row transforms,
    # AdmmScalars::skipa -- same mathematics, different rounding; one iteration more or less is 4 decades away)
    scale = float(a.abs().max())
    assert float((a - b).abs().max()) <= 2e-6 * scale
    rec.reset()
    rec._iterate(3)
    assert float((a - rec.get_image_estimate()[0]).abs().max()) >= 1e-3 * scale
    assert torch.isfinite(a).all() and float(a.min()) >= 0.0


def test_c2_channels_are_independent(c2):
    H, W, C, psf, g = c2
    y = torch.rand((H, W, C), device="cuda", generator=g)
    rgb = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
    rgb.set_data(y)
    full = rgb.apply(n_iter=3, disp_iter=None)
    gray = lpa.ADMM(psf[..., 1:2].contiguous(), tau=2e-6, mu2=1e-4)
    gray.set_data(y[..., 1:2].contiguous())
    one = gray.apply(n_iter=3, disp_iter=None)
    assert torch.equal(full[..., 1:2], one)                   # same kernels, same tiles: identical bits


def test_c5_depth_planes_match_single_plane_runs():
    D, H, W, C = 16, 1080, 1920, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    psf = torch.rand((D, H, W, C), device="cuda", generator=g) ** 12
    psf /= psf.norm()
    y = torch.rand((H, W, C), device="cuda", generator=g)
    rec = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
    assert rec._padded_shape == [16, 2160, 3840, 3]
    rec.set_data(y)
    full = rec.apply(n_iter=3, disp_iter=None)
    assert full.shape == (D, H, W, C)
    for d in (0, 7, 15):
        single = lpa.ADMM(psf[d:d + 1].contiguous(), tau=2e-6, mu2=1e-4)
        single.set_data(y)
        assert torch.equal(single.apply(n_iter=3, disp_iter=None)[0], full[d])


@pytest.mark.parametrize("shape", [(64, 8100, 1), (1000, 37, 3), (500, 500, 1), (333, 1025, 1)])
def test_unusual_shapes_against_oracle(shape):
    """Row length at the LDS limit (8100 -> 16200-point rows, 130 KB of LDS), odd padded widths (37 -> 75)
    with a four-step column split (1000 -> 2000 = 50 x 40), and non-power-of-two everything."""
    torch.set_num_threads(16)
    H, W, C = shape
    rng = np.random.default_rng(H + W)
    psf = orc.synthetic_psf(1, H, W, C, seed=H)
    y = rng.random((H, W, C), dtype=np.float32)
    x = torch.from_numpy(rng.standard_normal((1, 1, H, W, C)).astype(np.float32))
    cv = lpa.RealFFTConvolve2D(torch.from_numpy(psf).cuda(), pad=True)
    oc = orc.ConvolverOracle(psf, pad=True)
    assert cv._padded_shape[1:3] == [oc.geom.hp, oc.geom.wp]
    assert rel(cv.convolve(x.cuda()), oc.convolve(x)) <= 5e-6
    assert rel(cv.deconvolve(x.cuda()), oc.deconvolve(x)) <= 5e-6
    rec = lpa.ADMM(torch.from_numpy(psf).cuda(), tau=2e-6, mu2=1e-4)
    # none of these shapes is on anybody's list: each gets its compile-time-plan kernels from a plan module built on
    # first use (no silent drop to the run-time plans), and wherever the padded width is a multiple of 4 the X half, the
    # xi window and the H V skip follow
    info = rec._handle.plan_info()
    assert "plan module" in info and "[static" in info, info
    if rec._padded_shape[2] % 4 == 0:
        assert "row transforms skipped" in info, info
    rec.set_data(torch.from_numpy(y).cuda())
    o64 = orc.ADMMOracle(psf, dtype=torch.float64, tau=2e-6, mu2=1e-4)
    o64.set_data(y)
    assert rel(rec.apply(n_iter=4, disp_iter=None), o64.apply(4)) <= 1e-5
    f = lpa.FISTA(torch.from_numpy(psf).cuda())
    assert "plan module" in f._handle.plan_info() or rec._padded_shape[2] % 2 == 1, f._handle.plan_info()
    f.set_data(torch.from_numpy(y).cuda())
    of = orc.GDOracle(psf, kind="fista", dtype=torch.float64)
    of.set_data(y)
    assert rel(f.apply(n_iter=6, disp_iter=None), of.apply(6)) <= 1e-5


def test_too_wide_frame_is_rejected_cleanly():
    from lenslesspicam_amd._native import NativeError

    with pytest.raises(NativeError, match="16384"):
        lpa.RealFFTConvolve2D(torch.zeros((1, 8, 9000, 1), device="cuda"), pad=True)


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes_get_plan_modules_and_match_the_oracle(seed):
    """The reference accepts any (H, W) (rfft_convolve.py:110-117); so must the fast path.  Random frame shapes -- odd and
    even sizes, gray and colour, one or two depth planes -- each get their plan module on first use (never seen before:
    compiled here), and ADMM (TV-active, 6 iterations in one call: two of them on the sensor-window fast path where the
    padded width allows it), FISTA and the operator pair must match the float64 oracle."""
    torch.set_num_threads(16)
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(130, 900)), int(rng.integers(130, 1400))
    C, D = int(rng.choice([1, 3])), int(rng.choice([1, 1, 2]))
    psf = orc.synthetic_psf(D, H, W, C, seed=seed)
    y = rng.random((H, W, C), dtype=np.float32)
    kw = dict(tau=2e-6, mu2=1e-4)
    rec = lpa.ADMM(torch.from_numpy(psf).cuda(), **kw)
    info = rec._handle.plan_info()
    assert "plan module" in info and "[static" in info, (H, W, C, D, info)
    rec.set_data(torch.from_numpy(y).cuda())
    got = rec.apply(n_iter=6, disp_iter=None)
    for d in range(D):                                         # the reference's ADMM is 2-D: plane d alone
        o = orc.ADMMOracle(psf[d:d + 1], dtype=torch.float64, **kw)
        o.set_data(y)
        assert rel(got[d], o.apply(6)[0]) <= 1e-5, (H, W, C, D, d, info)
    f = lpa.FISTA(torch.from_numpy(psf).cuda())
    finfo = f._handle.plan_info()
    assert "plan module" in finfo or f._padded_shape[2] % 2 == 1, (H, W, finfo)
    f.set_data(torch.from_numpy(y).cuda())
    of = orc.GDOracle(psf, kind="fista", dtype=torch.float64)
    of.set_data(y)
    assert rel(f.apply(n_iter=6, disp_iter=None), of.apply(6)) <= 1e-5, (H, W, C, D, finfo)
    cv = lpa.RealFFTConvolve2D(torch.from_numpy(psf).cuda(), pad=True)
    oc = orc.ConvolverOracle(psf, pad=True)
    x = torch.from_numpy(rng.standard_normal((1, D, H, W, C)).astype(np.float32))
    assert rel(cv.convolve(x.cuda()), oc.convolve(x)) <= 5e-6 and rel(cv.deconvolve(x.cuda()), oc.deconvolve(x)) <= 5e-6


def test_no_compiler_on_the_gpu(tmp_path):
    """A deployment box without hipcc: a shape whose module is not on disk runs on the run-time
    plans of the core library -- identical results to float32 round-off, `plan_info` says so and names the missing module,
    one warning per missing module."""
    import warnings

    from lenslesspicam_amd import _native

    torch.set_num_threads(16)
    H, W, C = 333, 517, 3                          # on nobody's list
    rng = np.random.default_rng(77)
    psf = orc.synthetic_psf(1, H, W, C, seed=7)
    y = rng.random((H, W, C), dtype=np.float32)
    kw = dict(tau=2e-6, mu2=1e-4)
    opts = {"compiler": "/nonexistent/hipcc", "module_dir": str(tmp_path / "none")}
    _native._warned.clear()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        rec = lpa.ADMM(torch.from_numpy(psf).cuda(), engine_options=opts, **kw)
        fis = lpa.FISTA(torch.from_numpy(psf).cuda(), engine_options=opts)
        rec_again = lpa.ADMM(torch.from_numpy(psf).cuda(), engine_options=opts, **kw)
    info = rec._handle.plan_info()
    assert "run-time plans (" in info and "no hipcc" in info and "plan module" not in info, info
    assert rec_again._handle.fallback_reason() and fis._handle.fallback_reason()
    mine = [str(w.message) for w in caught if "run-time plans" in str(w.message)]
    # one warning per missing module (the message names it: what a deployment without a compiler would have to ship) --
    # ADMM's, reported once for the two ADMM solvers, and the gradient-descent family's
    assert len(mine) == 2 and "module f32_admm_" in mine[0] and "module f32_gd_" in mine[1], mine
    rec.set_data(torch.from_numpy(y).cuda())
    got = rec.apply(n_iter=6, disp_iter=None)
    with_module = lpa.ADMM(torch.from_numpy(psf).cuda(), engine_options={"module_dir": str(tmp_path / "mods")}, **kw)
    assert "plan module" in with_module._handle.plan_info()
    with_module.set_data(torch.from_numpy(y).cuda())
    want = with_module.apply(n_iter=6, disp_iter=None)
    assert rel(got, want) <= 3e-6
    o = orc.ADMMOracle(psf, dtype=torch.float64, **kw)
    o.set_data(y)
    assert rel(got, o.apply(6)) <= 1e-5
    fis.set_data(torch.from_numpy(y).cuda())
    of = orc.GDOracle(psf, kind="fista", dtype=torch.float64)
    of.set_data(y)
    assert rel(fis.apply(n_iter=6, disp_iter=None), of.apply(6)) <= 1e-5


def test_modules_unload_on_the_gpu(tmp_path):
    """option module_loaded_max: HIP plan modules nobody uses are dlclose()d (their code objects unregistered) and load
    again from disk when the shape returns -- bit-identical results, other handles unaffected."""
    torch.set_num_threads(16)
    opts = {"module_dir": str(tmp_path / "m"), "module_loaded_max": 1}
    rng = np.random.default_rng(5)

    def run(h, w, keep=False):
        psf = torch.from_numpy(orc.synthetic_psf(1, h, w, 1, seed=h)).cuda()
        y = torch.from_numpy(rng.random((h, w, 1), dtype=np.float32)).cuda()
        rec = lpa.ADMM(psf, tau=2e-6, mu2=1e-4, engine_options=opts)
        assert "plan module" in rec._handle.plan_info()
        rec.set_data(y)
        out = rec.apply(n_iter=3, disp_iter=None).clone()
        torch.cuda.synchronize()
        if not keep:
            rec._handle.close()
        return psf, y, out, rec

    a = run(300, 400)
    b = run(310, 410, keep=True)                  # stays referenced: must survive the unloading around it
    run(320, 420)
    run(330, 430)
    again = lpa.ADMM(a[0], tau=2e-6, mu2=1e-4, engine_options=opts)
    again.set_data(a[1])
    assert torch.equal(again.apply(n_iter=3, disp_iter=None), a[2])
    assert torch.equal(b[3].apply(n_iter=3, disp_iter=None), b[2])


@pytest.mark.parametrize("seed", range(4))
def test_random_batches_through_the_sequential_middle(seed):
    """Batches of small frames of random shape (single-pass columns of any 5-smooth length, 8 image columns per tile on
    256 / 512 / 1024 lanes, the 64-VGPR launch bound, loads up front for deep launches): the one-spectrum-at-a-time middle
    of a batch (k_cols_mid_admm_seq on the shape's own radices, compiled here on first use) against the per-frame float64
    oracle, and batch == single frames bit for bit on the same launch plan."""
    torch.set_num_threads(16)
    rng = np.random.default_rng(2000 + seed)
    # 512 < padded rows <= 1024 and enough (frame, plane, tile) workgroups: the shapes the engine gives that middle
    H, W = int(rng.integers(258, 500)), int(rng.integers(180, 420))
    C, B = 3, int(rng.choice([16, 24, 40]))
    psf = orc.synthetic_psf(1, H, W, C, seed=seed)
    frames = rng.random((B, H, W, C), dtype=np.float32)
    kw = dict(tau=2e-6, mu2=1e-4)
    rec = lpa.ADMM(torch.from_numpy(psf).cuda(), **kw)
    rec.set_data(torch.from_numpy(frames).cuda()[:, None])
    full = rec.apply_batch(n_iter=7)
    info = rec._handle.plan_info()
    assert "one spectrum at a time" in info, (H, W, C, B, info)
    for b in (0, B // 2, B - 1):
        o = orc.ADMMOracle(psf, dtype=torch.float64, **kw)
        o.set_data(frames[b])
        assert rel(full[b], o.apply(7)) <= 1e-5, (H, W, C, B, b, info)
    if "one spectrum at a time" in info:
        # (the same kernels: a single small frame would otherwise take the three-launch plan, option k1_rows)
        opts = {"mid_seq": 1, "mid_pre": int(info.rstrip().endswith("p")), "k1_rows": int("three launches" in info)}
        if "128 threads" in info.split("columns:")[0]:
            opts["prow_nt128"] = 1
        single = lpa.ADMM(torch.from_numpy(psf).cuda(), engine_options=opts, **kw)
        if single._handle.plan_info().split("plan module")[1] == info.split("plan module")[1]:
            for b in (0, B - 1):
                single.set_data(torch.from_numpy(frames[b]).cuda())
                assert torch.equal(single.apply(n_iter=7, disp_iter=None), full[b]), (H, W, C, B, b, info)
