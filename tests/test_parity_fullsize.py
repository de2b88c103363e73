"""
What is checked at BASELINE.json's sizes BESIDES the pins against the reference's own runs (tests/test_longrun_pins.py: every
config at its own size and length against samples the imported reference produced in float64 and float32) -- GPU only,
engine against engine, no CPU solver at 12 MP:

  C2  3040x4056x3 ADMM, 100 iterations in ONE call (97 of them on the steady-state path: xi inside the sensor window only,
      H V row transforms skipped outside it), default AND TV-active parameters, against (i) the same engine with that
      structure switched off (options hv_full, xi_full) and (ii) the float64 build with and without it (the structure is
      exact in real arithmetic: 1e-14 in float64)
  C5  the whole 16 x 1080x1920x3 stack, 50 iterations, TV-active parameters (the pins run the defaults): float32 vs the
      float64 build on every plane
  C4  batch of 64 DiffuserCam-sized frames (270x480x3), ADMM 20 iterations: 4 frames vs per-frame oracle apply(),
      all 64 bitwise vs single-frame runs (test/test_algos.py:198-229: batch == singles), and the same batch through
      lenslesspicam_amd.dist.reconstruct_sharded on an RCCL ("nccl") process group of world size 1
  torchrun  bench.py (headline and --config c4) launched the way the driver launches it for N > 1, with one rank
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import longrun_inputs as li  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="module")
def c2_inputs():
    """closed-form 12-MP inputs (tests/golden/longrun_inputs.py): sparse caustic-like PSF, a measurement-like frame, and
    the scene PSNR is quoted against"""
    H, W, C = 3040, 4056, 3
    return li.psf12(1, H, W, C, seed=0), li.scene(H, W, C), li.measurement(H, W, C, seed=0)


PLAIN = {"hv_full": 1, "xi_full": 1}     # launch plan without the sensor-window structure (include/lpc.h)


@pytest.fixture(scope="module")
def c2_tv_params(c2_inputs):
    """TV-active hyper-parameters at 12 MP.  The soft-threshold branch must be LIVE within 5 iterations: with a
    unit-energy 12-MP PSF the estimate is ~1e-3 and its finite differences ~1e-6, far below the default threshold
    tau/mu2 = 10 (and below the 0.02 that is enough at 270x480).  Take the largest tau of a decade ladder for which a
    sizeable part of U is non-zero but not all of it (decided on the engine, which costs milliseconds; the oracle then
    soft threshold of the fixture run is confirmed live by the reference, longrun_c2tv.npz)."""
    psf, _, y = c2_inputs
    psf_d, y_d = torch.from_numpy(psf).cuda(), torch.from_numpy(y).cuda()
    for tau in (2e-6, 2e-7, 2e-8, 2e-9, 2e-10, 2e-11, 2e-12):
        probe = lpa.ADMM(psf_d, tau=tau, mu2=1e-4)
        probe.set_data(y_d)
        probe._iterate(5)
        frac = float((probe._U != 0).float().mean())
        del probe
        torch.cuda.empty_cache()
        if frac > 0.05:
            print(f"TV-active parameters at 12 MP: tau={tau}, mu2=1e-4: {100 * frac:.1f} % of U non-zero after 5 iterations")
            return dict(tau=tau, mu2=1e-4)
    raise AssertionError("no threshold on the ladder activates the TV prox")


@pytest.mark.parametrize("tv_active", [True, False], ids=["tv_active", "defaults"])
def test_c2_admm_100_iterations_in_one_call(c2_inputs, c2_tv_params, tv_active):
    """BASELINE.json's headline is 100 iterations; inside one lpc_iterate() call iterations 2 ... 97 run with the sensor-
    window structure (AdmmScalars::skipa / skiphv / xiw), the last three complete.  Compare that call with (i) the same
    engine with the structure off (<= 5e-5 of max|ref|) and (ii) the float64 build with the structure off (<= 3e-4;
    and the float64 build with the structure ON agrees with it to 1e-9), <= 0.01 dB of PSNR vs the scene
    (north_star: "PSNR within 0.01 dB of reference" on 100-iteration ADMM at 4056x3040x3; reference loop
    lensless/recon/recon.py:575-576 over admm.py:313-338)."""
    psf, scene, y = c2_inputs
    psf_d, y_d = torch.from_numpy(psf).cuda(), torch.from_numpy(y).cuda()
    kw = c2_tv_params if tv_active else {}

    def run(**extra):
        rec = lpa.ADMM(psf_d.double() if extra.get("dtype") == "float64" else psf_d, **kw, **extra)
        info = rec._handle.plan_info()
        rec.set_data(y_d.double() if extra.get("dtype") == "float64" else y_d)
        out = rec.apply(n_iter=100, disp_iter=None)
        nz = float((rec._U != 0).float().mean()) if tv_active and not extra else None
        del rec
        torch.cuda.empty_cache()
        return out.cpu().numpy(), info, nz

    got, info, nz = run()
    assert "H V row transforms skipped" in info and "plan module" in info, info
    if tv_active:
        assert 0.02 < nz < 0.999, nz                           # the soft-threshold branch is still live after 100
    full, info_full, _ = run(engine_options=PLAIN)
    assert "xi inside the sensor window" not in info_full, info_full
    f64, info64, _ = run(dtype="float64", engine_options=PLAIN)
    assert "xi inside the sensor window" not in info64, info64
    # the structure is exact in real arithmetic: in float64 the call with it and the call without it agree to ~1e-12
    f64s, info64s, _ = run(dtype="float64")
    assert "H V row transforms skipped" in info64s, info64s
    e_id = rel(f64s, f64)
    print(f"C2 ADMM-100 float64, window structure on vs off: {e_id:.2e}")
    assert e_id <= 1e-9, e_id
    del f64s
    p = {k: orc.psnr(v[0].astype(np.float32), scene) for k, v in (("got", got), ("full", full), ("f64", f64))}
    e_full, e_64 = rel(got, full), rel(got, f64)
    print(f"C2 ADMM-100 {kw or 'defaults'}: vs structure-off {e_full:.2e} ({p['got'] - p['full']:+.2e} dB), "
          f"vs float64 build {e_64:.2e} ({p['got'] - p['f64']:+.2e} dB); PSNR vs scene {p['got']:.3f} dB")
    # The sensor-window structure is exact in real arithmetic: with it on / off the 100th iterate agrees to float32
    # round-off (measured 1.3e-5).  Against float64 TRUTH the float32 engine has drifted ~1e-4 of max|x| after 100
    # iterations (measured 9.3e-5 TV-active) -- rounding noise of 100 un-damped iterations, the same with the structure
    # off, and less than the reference's own float32 run (tests/golden/longrun_c2.npz, asserted by test_longrun_pins.py): bound
    # 3e-4, and the north_star criterion, PSNR vs the scene within 0.01 dB, on top.
    assert e_full <= 5e-5 and e_64 <= 3e-4, (e_full, e_64)
    assert abs(p["got"] - p["full"]) <= 0.01 and abs(p["got"] - p["f64"]) <= 0.01, p


def test_c5_all_16_planes_50_iterations_tv_active():
    """C5 at its own length and width with the soft threshold LIVE: ADMM 50 iterations of the whole 16 x 1080 x 1920 x 3
    stack in one call, float32 engine (half rows of 1920 points, 90 x 24 split, register middle, sensor-window structure
    for 46 of the 50) against the float64 build WITHOUT the window structure, every one of the 16 planes (the float64 build
    at this shape is pinned to the reference's own float64 run by tests/test_longrun_pins.py)."""
    D, H, W, C = 16, 1080, 1920, 3
    psf = np.concatenate([li.psf12(1, H, W, C, seed=d) for d in range(D)])
    scene, y = li.scene(H, W, C), li.measurement(H, W, C, seed=0)
    kw = dict(tau=2e-6, mu2=1e-4)
    r64 = lpa.ADMM(torch.from_numpy(psf).cuda().double(), dtype="float64", engine_options=PLAIN, **kw)
    r64.set_data(torch.from_numpy(y).cuda().double())
    t50 = r64.apply(n_iter=50, disp_iter=None).cpu().numpy()
    del r64
    torch.cuda.empty_cache()
    rec = lpa.ADMM(torch.from_numpy(psf).cuda(), **kw)
    info = rec._handle.plan_info()
    assert rec._padded_shape == [16, 2160, 3840, 3] and "H V row transforms skipped" in info and "plan module" in info, info
    rec.set_data(torch.from_numpy(y).cuda())
    g50 = rec.apply(n_iter=50, disp_iter=None).cpu().numpy()
    nz = float((rec._U != 0).float().mean())
    del rec
    torch.cuda.empty_cache()
    errs = [rel(g50[k], t50[k]) for k in range(D)]
    dps = [li.stats(g50[k], scene)[4] - li.stats(t50[k], scene)[4] for k in range(D)]
    print(f"C5 ADMM-50 TV-active, 16 planes: float32 vs float64 build max {max(errs):.2e} (plane {int(np.argmax(errs))}), "
          f"median {float(np.median(errs)):.2e}; |PSNR delta| max {max(abs(v) for v in dps):.2e} dB; U non-zero {100 * nz:.1f} %")
    assert 0.005 < nz < 0.999, nz          # the soft-threshold branch is live
    assert max(errs) <= 3e-4 and max(abs(v) for v in dps) <= 0.01, (errs, dps)


# ------------------------------------------------------------------------------------- C4 --
@pytest.fixture(scope="module")
def c4_inputs():
    torch.set_num_threads(16)
    B, H, W, C = 64, 270, 480, 3
    psf = orc.synthetic_psf(1, H, W, C, seed=0)
    frames = np.stack([orc.synthetic_measurement(psf, orc.synthetic_scene(H, W, C, seed=1 + b)) for b in range(B)])
    return psf, frames


def test_c4_batch64_vs_oracle_and_singles(c4_inputs):
    psf, frames = c4_inputs
    psf_d, frames_d = torch.from_numpy(psf).cuda(), torch.from_numpy(frames).cuda()
    rec = lpa.ADMM(psf_d)
    rec.set_data(frames_d[:, None])
    full = rec.apply_batch(n_iter=20)
    assert full.shape == (64, 1, 270, 480, 3)
    for b in (0, 21, 42, 63):                                  # per-frame oracle apply(), the reference's own loop
        o = orc.ADMMOracle(psf)
        o.set_data(frames[b])
        assert rel(full[b], o.apply(20)) <= 1e-5, b
    # batch == singles (test/test_algos.py:198-229).  Bit for bit when the single-frame solver runs the kernels the
    # batch ran: the engine picks the one-spectrum-at-a-time middle (with both tiles' loads issued up front) for large
    # batches only; the options mid_seq / mid_pre select it for one frame too (the rows are the same since round 5: 256
    # lanes with the TV / W half inside, at every batch size).  The default single-frame plan (two spectra side by side,
    # 8 columns each) is another instruction stream for the same arithmetic: equal to float32 round-off.
    single = lpa.ADMM(psf_d, engine_options={"mid_seq": 1, "mid_pre": 1})
    assert "three launches per iteration" in rec._handle.plan_info()
    assert "one spectrum at a time" in single._handle.plan_info() and "one spectrum at a time" in rec._handle.plan_info()
    assert "row transforms skipped" in single._handle.plan_info() and "row transforms skipped" in rec._handle.plan_info()
    assert single._handle.plan_info().split("plan module")[1] == rec._handle.plan_info().split("plan module")[1]
    for b in range(64):
        single.set_data(frames_d[b])
        assert torch.equal(single.apply(n_iter=20, disp_iter=None), full[b]), b
    plain = lpa.ADMM(psf_d)
    assert "one spectrum at a time" not in plain._handle.plan_info()
    for b in (0, 31, 63):
        plain.set_data(frames_d[b])
        assert rel(plain.apply(n_iter=20, disp_iter=None), full[b]) <= 2e-6, b


def test_c4_sharded_through_rccl_world1(c4_inputs):
    import torch.distributed as dist

    from lenslesspicam_amd.dist import ShardedReconstructor, reconstruct_sharded

    psf, frames = c4_inputs
    psf_d, frames_d = torch.from_numpy(psf).cuda(), torch.from_numpy(frames).cuda()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        ref = lpa.ADMM(psf_d)
        ref.set_data(frames_d[:, None])
        want = ref.apply_batch(n_iter=20)
        got = reconstruct_sharded(lpa.ADMM, psf_d, frames_d, n_iter=20)
        assert torch.equal(got, want)
        # the collective itself, on RCCL: world size 1 takes the early exit above, so drive the gather explicitly
        sr = ShardedReconstructor(lpa.ADMM, psf_d, solver=ref)
        send = want.contiguous()
        recv = torch.empty_like(send)
        dist.all_gather_into_tensor(recv, send)
        torch.cuda.synchronize()
        assert torch.equal(recv, want)
        five = lpa.ADMM(psf_d)
        five.set_data(frames_d[:5, None])
        assert torch.equal(sr(frames_d[:5], n_iter=3), five.apply_batch(n_iter=3))     # solver re-used, new batch size
    finally:
        if own:
            dist.destroy_process_group()


# ------------------------------------------------------------------- bench.py under torchrun --
def _torchrun_bench(extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_headline_under_torchrun_one_rank():
    r = _torchrun_bench(["--n-iter", "10", "--no-other-configs"], 29521)
    assert r["n_gpus"] == 1 and r["unit"] == "iterations/s" and r["value"] > 50
    assert r["roofline"]["bound"] == "hbm" and 0.2 < r["roofline"]["frac"] < 1.0


def test_bench_c4_under_torchrun_one_rank():
    r = _torchrun_bench(["--config", "c4"], 29523)
    assert r["n_gpus"] == 1 and r["unit"] == "frame-iterations/s" and r["value"] > 5000
