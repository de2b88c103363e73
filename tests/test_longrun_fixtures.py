"""
CPU side of the BASELINE-length pins (tests/golden/longrun_*.npz, gen_longrun.py): the closed-form inputs rebuild to the
generator's fingerprints, every fixture is self-consistent, and the kernel sources (SIMT emulator, same C ABI) reproduce
the reference's own C4 frame at its own length -- float32 and float64 -- from nothing but the committed numbers.
The 12-MP comparisons are tests/test_longrun_pins.py (-m gpu).
"""
import glob
import os
import sys

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import longrun_inputs as li  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_inputs_are_exact_operations_and_rebuild_to_the_stored_fingerprints():
    fx = np.load(os.path.join(GOLDEN, "longrun_c4.npz"))
    h, w, c = 270, 480, 3
    psf = li.psf12(1, h, w, c, seed=0)
    assert psf.dtype == np.float32 and abs(float(np.linalg.norm(psf.astype(np.float64))) - 1.0) < 5e-3   # unit energy
    for k in (0, 21, 42, 63):
        y = li.measurement(h, w, c, seed=k)
        assert y.dtype == np.float32 and float(y.max()) == 1.0 and float(y.min()) >= 0.0
        assert li.fingerprint(psf)[0] == fx[f"frame{k}_psf_fp"][0]
        assert li.fingerprint(y)[0] == fx[f"frame{k}_data_fp"][0]
    sc = li.scene(h, w, c)
    assert float(sc.max()) > 0 and (sc[:20] == 0).all() and (sc[:, :20] == 0).all()      # compact bumps, central 60 %


def test_every_fixture_is_self_consistent():
    files = sorted(glob.glob(os.path.join(GOLDEN, "longrun_*.npz")))
    assert files, "no longrun fixture committed"
    for path in files:
        fx = np.load(path)
        names = sorted({k[:-len("_iters")] for k in fx.files if k.endswith("_iters")})
        assert names, path
        for name in names:
            for it in fx[f"{name}_iters"]:
                c64, c32 = fx[f"{name}_f64_it{it}_crops"], fx[f"{name}_f32_it{it}_crops"]
                assert c64.dtype == np.float64 and c32.dtype == np.float32 and c64.shape == c32.shape
                assert c64.shape[0] == 8 and c64.shape[1:3] == (32, 32)
                top = fx[f"{name}_f64_it{it}_stats"][2]
                l64, l32 = fx[f"{name}_f64_it{it}_lattice"], fx[f"{name}_f32_it{it}_lattice"]
                assert l64.dtype == np.float64 and l32.dtype == np.float32 and l64.shape == l32.shape and l64.max() > 0
                # (a crop may be all zero -- FISTA's projection clamps most of the frame's border -- the lattice never is)
                d = max(np.abs(c32.astype(np.float64) - c64).max(), np.abs(l32.astype(np.float64) - l64).max()) / top
                full = float(fx[f"{name}_f32_it{it}_dist64_full"])
                assert 0 < d <= full * (1 + 1e-12), (path, name, it, d, full)     # the samples are part of the frame
                assert (c64 >= 0).all() and (c32 >= 0).all()                       # apply() clamps (admm.py:337, gd.py:27-30)
                assert abs(fx[f"{name}_f32_it{it}_stats"][4] - fx[f"{name}_f64_it{it}_stats"][4]) <= 0.01   # PSNR, dB


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_c4_frame_at_its_own_length_on_the_kernel_sources(backend, dtype):
    """Frame 21 of C4, ADMM 20 iterations, against the REFERENCE's own run (not the oracle)."""
    if backend.kind == "emu" and dtype == "float64" and backend.lib.f64 is None:
        pytest.skip("sanitizer flavour: float32 only")
    fx = np.load(os.path.join(GOLDEN, "longrun_c4.npz"))
    h, w, c = 270, 480, 3
    tdt = torch.float64 if dtype == "float64" else torch.float32
    psf, y = li.psf12(1, h, w, c, seed=0), li.measurement(h, w, c, seed=21)
    rec = lpa.ADMM(torch.from_numpy(psf).to(backend.device, tdt), dtype=dtype)
    rec.set_data(torch.from_numpy(y).to(backend.device, tdt))
    out = rec.apply(n_iter=20, disp_iter=None).cpu().numpy()[0]
    crops, lat = li.samples(out)
    top = fx["frame21_f64_it20_stats"][2]
    d = max(np.abs(crops - fx["frame21_f64_it20_crops"]).max(), np.abs(lat - fx["frame21_f64_it20_lattice"]).max()) / top
    s = li.stats(out, li.scene(h, w, c))
    print(f"C4 frame 21, {dtype}: vs the reference's float64 samples {d:.2e}, PSNR {s[4]:.5f} vs {fx['frame21_f64_it20_stats'][4]:.5f}")
    assert d <= (1e-9 if dtype == "float64" else 5e-6), d
    assert abs(s[4] - fx["frame21_f64_it20_stats"][4]) <= (1e-6 if dtype == "float64" else 0.01)
