"""
BASELINE.json's configurations at their OWN size and length against values the REFERENCE ITSELF produced
(tests/golden/longrun_*.npz, made once in the build container by tests/golden/gen_longrun.py: the imported reference in
torch-CPU float64 AND float32 on closed-form inputs, reduced to crops + a lattice + frame statistics):

  C2    3040x4056x3 ADMM, default parameters, 5 / 30 / 100 iterations   (recon.py:575-576 over admm.py:313-338)
  C2tv  the same frame with the soft threshold live, 5 / 30 / 100
  C3    the same frame, FISTA 6 / 30 / 300 iterations                   (gd.py:235-241)
  C5    planes 0 and 7 of the 16 x 1080x1920x3 stack, ADMM 12 / 50, run as the whole stack
  C4    frames 0, 21, 42, 63 of the batch of 64 DiffuserCam-sized frames, ADMM 20, run as the whole batch

What is asserted, per snapshot:
  * the float64 build (liblpc_f64.so, sensor-window structure ON -- the kernels the float32 engine runs) equals the
    reference's float64 samples to <= 1e-9 of max|x|, its frame sums to 1e-9, its PSNR vs the scene to 1e-6 dB;
  * the float32 engine is within SURVEY.md section 8(c)'s tolerances of the reference's float64 samples (TOL32 below) and
    within 0.01 dB of its PSNR (north_star);
  * the ATTRIBUTION behind the float32 tolerance: the engine is no further from the reference's float64 samples than the
    reference's own float32 run is (ADMM; FISTA: no further than twice that -- ATTR_SLACK; a floor of a few ulp where both
    are at round-off).

No oracle, no CPU solver: the inputs are rebuilt bit for bit (longrun_inputs.py, fingerprints checked) and everything
else is a few hundred KB of committed numbers -- zero host minutes next to the oracle-based anchors they replace.
"""
import os
import sys

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import longrun_inputs as li  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

F64_TOL = 1e-9
# engine distance <= ATTR_SLACK x the reference's own float32 distance, on the same samples.  ADMM multiplies by 1/mu1 = 1e6
# outside the sensor area, which amplifies FFT round-off: there the engine is 5-20 x CLOSER to float64 truth than torch's CPU
# FFT (1.0).  FISTA has no such amplifier: both float32 runs accumulate plain round-off of the same size (measured at 300
# iterations: 4.6e-5 against the reference's 4.1e-5 on the samples, 1.1e-4 over its full frame) -- "no worse than twice".
ATTR_SLACK = {"admm": 1.0, "fista": 2.0}
ATTR_FLOOR = 5e-6         # ... or this, where both sit at float32 round-off (short runs, small frames)
# float32 tolerance vs float64 truth on the samples, relative to max|x| -- SURVEY.md section 8(c)'s own figures (ADMM 1e-5 /
# 5e-5 after <= 20 / 100 iterations, FISTA 5e-4 after 300); measured: 1.1e-5 at ADMM-100 where the reference's own float32
# run is at 9.8e-5 (the full-frame bound of tests/test_parity_fullsize.py stays 3e-4: other inputs have shown 1.7e-4)
TOL32 = {("admm", 5): 1e-5, ("admm", 12): 1e-5, ("admm", 20): 1e-5, ("admm", 30): 2e-5, ("admm", 50): 2e-5,
         ("admm", 100): 5e-5, ("fista", 6): 1e-5, ("fista", 30): 5e-5, ("fista", 300): 5e-4}


def fixture_file(tag):
    path = os.path.join(GOLDEN, f"longrun_{tag}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated yet (tests/golden/gen_longrun.py {tag})")
    return np.load(path)


def check_inputs(fx, name, psf, data):
    for key, arr in ((f"{name}_psf_fp", psf), (f"{name}_data_fp", data)):
        want, got = fx[key], li.fingerprint(arr)
        assert want[0] == got[0], f"{key}: the closed-form inputs differ from the generator's (CRC {got[0]:.0f} vs {want[0]:.0f})"
        assert abs(want[1] - got[1]) <= 1e-9 * abs(want[1])


def sample_dist(img, crops_ref, lat_ref, top):
    crops, lat = li.samples(img)
    return max(float(np.abs(crops.astype(np.float64) - crops_ref).max()),
               float(np.abs(lat.astype(np.float64) - lat_ref).max())) / top


def compare(fx, name, it, kind, out32, out64, scene):
    """out32 / out64: (H, W, C) outputs of the float32 engine / the float64 build after `it` iterations."""
    r64c, r64l, r64s = (fx[f"{name}_f64_it{it}_{k}"] for k in ("crops", "lattice", "stats"))
    r32c, r32l, r32s = (fx[f"{name}_f32_it{it}_{k}"] for k in ("crops", "lattice", "stats"))
    top = r64s[2]
    # -- the float64 build against the reference's float64 run
    d64 = sample_dist(out64, r64c, r64l, top)
    s64 = li.stats(out64, scene)
    sums = max(abs(s64[0] - r64s[0]) / abs(r64s[0]), abs(s64[1] - r64s[1]) / abs(r64s[1]), abs(s64[2] - r64s[2]) / top)
    # -- the float32 engine against the same, next to the reference's own float32 run
    d32 = sample_dist(out32, r64c, r64l, top)
    dref = max(float(np.abs(r32c.astype(np.float64) - r64c).max()), float(np.abs(r32l.astype(np.float64) - r64l).max())) / top
    s32 = li.stats(out32, scene)
    print(f"{name} it {it}: float64 build vs reference float64 {d64:.2e} (sums {sums:.1e}, PSNR {s64[4] - r64s[4]:+.1e} dB); "
          f"float32 engine vs reference float64 {d32:.2e}, reference float32 vs float64 {dref:.2e} on the samples "
          f"({float(fx[f'{name}_f32_it{it}_dist64_full']):.2e} full frame); PSNR engine {s32[4]:.5f} ref64 {r64s[4]:.5f} "
          f"ref32 {r32s[4]:.5f} dB")
    assert d64 <= F64_TOL and sums <= F64_TOL and abs(s64[4] - r64s[4]) <= 1e-6, (name, it, d64, sums)
    assert d32 <= TOL32[(kind, it)], (name, it, d32)
    assert abs(s32[4] - r64s[4]) <= 0.01, (name, it, s32[4], r64s[4])
    assert d32 <= max(ATTR_SLACK[kind] * dref, ATTR_FLOOR), (name, it, d32, dref)


def run(cls, psf, data, n, dtype, batch=False, **kw):
    dev = torch.device("cuda", 0)
    tdt = torch.float64 if dtype == "float64" else torch.float32
    rec = cls(torch.from_numpy(psf).to(dev, tdt), dtype=dtype, **kw)
    y = torch.from_numpy(data).to(dev, tdt)
    rec.set_data(y[:, None] if batch else y)
    out = rec.apply_batch(n_iter=n) if batch else rec.apply(n_iter=n, disp_iter=None)
    info = rec._handle.plan_info()
    del rec
    torch.cuda.empty_cache()
    return out.cpu().numpy(), info


def params_kw(fx, name):
    p = fx[f"{name}_f64_params"]
    return dict(mu1=float(p[0]), mu2=float(p[1]), mu3=float(p[2]), tau=float(p[3]))


# ------------------------------------------------------------------------------------------------------- C4 --
def test_c4_batch_of_64_against_the_reference():
    fx = fixture_file("c4")
    h, w, c = 270, 480, 3
    psf, scene = li.psf12(1, h, w, c, seed=0), li.scene(h, w, c)
    frames = np.stack([li.measurement(h, w, c, seed=k) for k in range(64)])
    for k in (0, 21, 42, 63):
        check_inputs(fx, f"frame{k}", psf, frames[k])
    o32, info = run(lpa.ADMM, psf, frames, 20, "float32", batch=True)
    assert "one spectrum at a time" in info and "three launches per iteration" in info, info
    o64, _ = run(lpa.ADMM, psf, frames, 20, "float64", batch=True)
    for k in (0, 21, 42, 63):
        compare(fx, f"frame{k}", 20, "admm", o32[k, 0], o64[k, 0], scene)


# ------------------------------------------------------------------------------------------------------- C5 --
def test_c5_depth_stack_planes_against_the_reference():
    fx = fixture_file("c5")
    h, w, c = 1080, 1920, 3
    data, scene = li.measurement(h, w, c, seed=0), li.scene(h, w, c)
    psf = np.concatenate([li.psf12(1, h, w, c, seed=d) for d in range(16)])
    for d in (0, 7):
        check_inputs(fx, f"plane{d}", psf[d:d + 1], data)
    for it in (12, 50):
        o32, info = run(lpa.ADMM, psf, data, it, "float32")
        assert o32.shape == (16, h, w, c) and "plan module" in info, info
        o64, _ = run(lpa.ADMM, psf, data, it, "float64")
        for d in (0, 7):
            compare(fx, f"plane{d}", it, "admm", o32[d], o64[d], scene)


# -------------------------------------------------------------------------------------------------- C2 / C3 --
@pytest.fixture(scope="module")
def c2():
    h, w, c = 3040, 4056, 3
    return li.psf12(1, h, w, c, seed=0), li.measurement(h, w, c, seed=0), li.scene(h, w, c)


@pytest.mark.parametrize("tag,name", [("c2", "admm"), ("c2tv", "admm_tv")], ids=["defaults", "tv_active"])
def test_c2_admm_100_iterations_against_the_reference(c2, tag, name):
    fx = fixture_file(tag)
    psf, data, scene = c2
    check_inputs(fx, name, psf, data)
    kw = params_kw(fx, name)
    if tag == "c2tv":
        assert 0.02 < float(fx[f"{name}_f64_U_nonzero"]) < 0.999        # the reference's soft threshold is live after 100
    for it in (int(v) for v in fx[f"{name}_iters"]):
        o32, info = run(lpa.ADMM, psf, data, it, "float32", **kw)
        if it >= 5:
            assert "H V row transforms skipped" in info and "plan module" in info, info
        o64, info64 = run(lpa.ADMM, psf, data, it, "float64", **kw)
        assert "H V row transforms skipped" in info64, info64
        compare(fx, name, it, "admm", o32[0], o64[0], scene)


def test_c3_fista_300_iterations_against_the_reference(c2):
    fx = fixture_file("c3")
    psf, data, scene = c2
    check_inputs(fx, "fista", psf, data)
    for it in (int(v) for v in fx["fista_iters"]):
        o32, info = run(lpa.FISTA, psf, data, it, "float32")
        assert "plan module" in info, info
        o64, _ = run(lpa.FISTA, psf, data, it, "float64")
        compare(fx, "fista", it, "fista", o32[0], o64[0], scene)
