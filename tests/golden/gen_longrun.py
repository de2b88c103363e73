#!/usr/bin/env python3
"""
BASELINE.json's configs AT THEIR OWN SIZE AND LENGTH, run once through the REAL reference (read-only mount at
/root/reference, torch-CPU, float64 AND float32), reduced to a few hundred KB of samples per config:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_longrun.py [c4 c5 c2 c3 c2tv]      (hours of CPU; build container)

  c2    3040x4056x3  ADMM (default parameters)   snapshots after 5, 30, 100 iterations  (recon.py:575-576 over admm.py:313-338)
  c2tv  the same frame, soft threshold live      snapshots after 5, 30, 100
  c3    3040x4056x3  FISTA                       snapshots after 6, 30, 300             (gd.py:235-241)
  c5    planes 0 and 7 of the 16 x 1080x1920x3 stack, ADMM, 12 and 50 iterations (plane d alone)
  c4    frames 0, 21, 42, 63 of the batch of 64 DiffuserCam-sized frames (270x480x3), ADMM 20 iterations

Inputs are closed-form (longrun_inputs.py) so the GPU box rebuilds the same bits; their fingerprints are stored.
Per snapshot and precision: 8 crops of 32x32xC, a stride-61 lattice over the whole frame, [sum, sum^2, max, min, PSNR vs
the scene], and -- float32 only -- the full-frame distance max|ref32 - ref64| / max|ref64| (the yardstick the
float32 tolerance is attributed to).  Outputs: tests/golden/longrun_<config>.npz.  No reference source is stored.
"""
import gc
import os
import sys
import time
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", MagicMock())
sys.path.insert(0, os.environ.get("LENSLESS_REFERENCE", "/root/reference"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from lensless.recon.admm import ADMM  # noqa: E402
from lensless.recon.gd import FISTA  # noqa: E402

import longrun_inputs as li  # noqa: E402

OUT = os.environ.get("LONGRUN_OUT", os.path.dirname(os.path.abspath(__file__)))
TMP = os.environ.get("LONGRUN_TMP", "/tmp/longrun")
os.makedirs(TMP, exist_ok=True)
torch.set_num_threads(int(os.environ.get("LONGRUN_THREADS", "6")))


def snapshot(rec, kind):
    """What apply() would return now, WITHOUT the in-place clamp of ADMM._form_image (admm.py:332-338) so that the
    trajectory is the one of a single apply(n_iter=max) call."""
    if kind == "admm":
        return rec._convolver._crop(rec._image_est).clamp(min=0)[0, 0].numpy().copy()
    return rec._form_image()[0, 0].numpy().copy()          # GD family: proj() returns a new tensor


def run(kind, psf, data, iters, dtype, **kw):
    tdt = torch.float64 if dtype == "float64" else torch.float32
    cls = ADMM if kind == "admm" else FISTA
    rec = cls(torch.from_numpy(psf).to(tdt), dtype=dtype, **kw)
    rec.set_data(torch.from_numpy(data).to(tdt))
    rec.reset()
    snaps = {}
    t0 = time.time()
    for i in range(max(iters)):
        rec._update(i)
        if (i + 1) in iters:
            snaps[i + 1] = snapshot(rec, kind)
            print(f"    {kind} {dtype} it {i + 1}: {time.time() - t0:.0f} s", flush=True)
    final = rec._form_image()[0][0].numpy()
    assert np.array_equal(final, snaps[max(iters)])          # apply()'s return value == the last snapshot
    extra = {}
    if kind == "admm":
        extra["U_nonzero"] = float((rec._U != 0).double().mean())
        extra["params"] = np.array([rec._mu1, rec._mu2, rec._mu3, rec._tau], dtype=np.float64)
    return snaps, extra


def case(name, kind, psf, data, scene, iters, out, **kw):
    """Run float64 then float32; add the samples of every snapshot to ``out`` under '<name>_...'."""
    ref64, ex = run(kind, psf, data, iters, "float64", **kw)
    for k, v in ex.items():
        out[f"{name}_f64_{k}"] = v
    for it, img in ref64.items():
        crops, lat = li.samples(img)
        out[f"{name}_f64_it{it}_crops"], out[f"{name}_f64_it{it}_lattice"] = crops, lat
        out[f"{name}_f64_it{it}_stats"] = li.stats(img, scene)
    gc.collect()
    ref32, ex = run(kind, psf, data, iters, "float32", **kw)
    for k, v in ex.items():
        out[f"{name}_f32_{k}"] = v
    for it, img in ref32.items():
        crops, lat = li.samples(img)
        out[f"{name}_f32_it{it}_crops"], out[f"{name}_f32_it{it}_lattice"] = crops, lat
        out[f"{name}_f32_it{it}_stats"] = li.stats(img, scene)
        d = np.abs(img.astype(np.float64) - ref64[it]).max() / np.abs(ref64[it]).max()
        out[f"{name}_f32_it{it}_dist64_full"] = d
        print(f"  {name} it {it}: reference float32 vs float64, full frame {d:.3e}; "
              f"PSNR {out[f'{name}_f32_it{it}_stats'][4]:.5f} vs {out[f'{name}_f64_it{it}_stats'][4]:.5f}", flush=True)
    out[f"{name}_iters"] = np.array(sorted(iters))
    out[f"{name}_psf_fp"] = li.fingerprint(psf)
    out[f"{name}_data_fp"] = li.fingerprint(data)


def save(tag, out):
    np.savez_compressed(os.path.join(OUT, f"longrun_{tag}.npz"), **out)
    print("wrote", tag, f"{os.path.getsize(os.path.join(OUT, f'longrun_{tag}.npz')) / 1e3:.0f} KB", flush=True)


def gen_c4():
    h, w, c = 270, 480, 3
    psf, sc, out = li.psf12(1, h, w, c, seed=0), li.scene(h, w, c), {}
    for k in (0, 21, 42, 63):
        case(f"frame{k}", "admm", psf, li.measurement(h, w, c, seed=k), sc, [20], out)
    save("c4", out)


def gen_c5():
    h, w, c = 1080, 1920, 3
    data, sc, out = li.measurement(h, w, c, seed=0), li.scene(h, w, c), {}
    for d in (0, 7):
        case(f"plane{d}", "admm", li.psf12(1, h, w, c, seed=d), data, sc, [12, 50], out)
    save("c5", out)


def c2_inputs():
    h, w, c = 3040, 4056, 3
    return li.psf12(1, h, w, c, seed=0), li.measurement(h, w, c, seed=0), li.scene(h, w, c)


def gen_c2():
    psf, data, sc = c2_inputs()
    out = {}
    case("admm", "admm", psf, data, sc, [5, 30, 100], out)
    save("c2", out)


def gen_c3():
    psf, data, sc = c2_inputs()
    out = {}
    case("fista", "fista", psf, data, sc, [6, 30, 300], out)
    save("c3", out)


def gen_c2tv():
    """Soft threshold LIVE at 12 MP: the largest tau of a decade ladder (mu2 = 1e-4) for which, after 5 float32
    iterations of the reference, between 5 % and 95 % of U is non-zero."""
    psf, data, sc = c2_inputs()
    chosen = None
    # (the ladder as it ran: 2e-6, 2e-7 -> 0 %, 2e-8 -> 2.4 %, 2e-9 -> 45.6 % of U non-zero; LONGRUN_TV_FROM=2e-9 resumes there)
    first = float(os.environ.get("LONGRUN_TV_FROM", "1"))
    for tau in (2e-6, 2e-7, 2e-8, 2e-9, 2e-10, 2e-11, 2e-12):
        if tau > first * 1.0000001:
            continue
        _, ex = run("admm", psf, data, [5], "float32", tau=tau, mu2=1e-4)
        print(f"  tau={tau:g}: {100 * ex['U_nonzero']:.1f} % of U non-zero after 5 iterations", flush=True)
        gc.collect()
        if 0.05 < ex["U_nonzero"] < 0.95:
            chosen = tau
            break
    assert chosen is not None
    out = {}
    case("admm_tv", "admm", psf, data, sc, [5, 30, 100], out, tau=chosen, mu2=1e-4)
    save("c2tv", out)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["c4", "c5", "c2", "c3", "c2tv"]
    for t in todo:
        t0 = time.time()
        print("==", t, flush=True)
        {"c4": gen_c4, "c5": gen_c5, "c2": gen_c2, "c3": gen_c3, "c2tv": gen_c2tv}[t]()
        print(f"== {t} done in {(time.time() - t0) / 60:.1f} min", flush=True)
