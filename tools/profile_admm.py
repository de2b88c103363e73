#!/usr/bin/env python3
"""
Counterpart of the reference's timing harness ``profile/admm.py`` (and, with --algo fista, of
``profile/gradient_descent.py``): same protocol -- build the solver, one warm-up ``apply``, then
``n_trials`` timed ``apply`` + ``reset`` pairs -- but with a device synchronisation inside the timed
region (the reference's GPU leg has none, profile/admm.py:98-108, so it times enqueueing).

The reference feeds ``load_data(psf_fp, data_fp, downsample=4, gray=True)``; its PNGs are not
redistributable here, so the frame is synthetic at the size that setting produces for the RPi HQ
sensor (3040 x 4056 / 4 = 760 x 1014, gray).  Pass --psf/--data .npy files (already preprocessed,
shapes (D,H,W,C) and (H,W,C)) to profile real captures.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lenslesspicam_amd as lpa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="admm", choices=["admm", "fista", "nesterov", "gd"])
    ap.add_argument("--n-iter", type=int, default=None, help="default: 5 (admm) / 300 (gd family), like the reference")
    ap.add_argument("--n-trials", type=int, default=None, help="default: 10 (admm) / 3 (gd family)")
    ap.add_argument("--height", type=int, default=760)
    ap.add_argument("--width", type=int, default=1014)
    ap.add_argument("--rgb", action="store_true", help="reference default is gray=True")
    ap.add_argument("--psf")
    ap.add_argument("--data")
    ap.add_argument("--raw", action="store_true",
                    help="the reference's own plumbing from a RAW capture: synthetic 12-bit 3040 x 4056 x 3 PSF and frame "
                         "written as .npy, then load_data(psf_fp, data_fp, downsample=4, gray=True) like profile/admm.py:19-26")
    args = ap.parse_args()
    n_iter = args.n_iter or (5 if args.algo == "admm" else 300)
    n_trials = args.n_trials or (10 if args.algo == "admm" else 3)
    dev = torch.device("cuda")
    if args.raw:
        import tempfile

        from lenslesspicam_amd.prep import load_data

        rng = np.random.default_rng(0)
        tmp = tempfile.mkdtemp()
        H0, W0 = 3040, 4056
        raw_psf = (rng.random((1, H0, W0, 3), dtype=np.float32) ** 12 * 3800 + 64).astype(np.uint16)
        raw_dat = (rng.random((H0, W0, 3), dtype=np.float32) * 3500 + 64).astype(np.uint16)
        np.save(os.path.join(tmp, "psf.npy"), raw_psf)
        np.save(os.path.join(tmp, "raw.npy"), raw_dat)
        torch.cuda.synchronize()
        t0 = time.time()
        psf, data = load_data(os.path.join(tmp, "psf.npy"), os.path.join(tmp, "raw.npy"), downsample=4, plot=False,
                              gray=True, dtype="float32", use_torch=True, torch_device="cuda", bgr_input=False)
        torch.cuda.synchronize()
        print(f"load_data(downsample=4, gray=True) from 2 x {raw_psf.nbytes / 1e6:.0f} MB .npy files: "
              f"{(time.time() - t0) * 1e3:.1f} ms (file read + upload + device preparation and resize) -> {tuple(psf.shape)}")
        data = data[0]
    elif args.psf:
        psf = torch.from_numpy(np.load(args.psf).astype(np.float32)).to(dev)
        data = torch.from_numpy(np.load(args.data).astype(np.float32)).to(dev)
    else:
        C = 3 if args.rgb else 1
        g = torch.Generator(device="cuda").manual_seed(0)
        psf = torch.rand((1, args.height, args.width, C), device=dev, generator=g) ** 12
        psf /= psf.norm()                                   # lensless/utils/io.py:375
        data = torch.rand((args.height, args.width, C), device=dev, generator=g)
        data /= data.max()                                  # lensless/utils/io.py:196-197
    cls = {"admm": lpa.ADMM, "fista": lpa.FISTA, "nesterov": lpa.NesterovGradientDescent,
           "gd": lpa.GradientDescent}[args.algo]
    recon = cls(psf, dtype="float32")
    recon.set_data(data)
    recon.apply(n_iter=n_iter, disp_iter=None, plot=False)   # warm-up, like profile/admm.py:35
    recon.reset()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(n_trials):
        start = time.time()
        recon.apply(n_iter=n_iter, disp_iter=None, plot=False)
        torch.cuda.synchronize()
        total += time.time() - start
        recon.reset()
    avg = total / n_trials
    print(f"lenslesspicam_amd {cls.__name__} on {tuple(psf.shape)}: {avg * 1e3:.3f} ms per {n_iter}-iteration apply "
          f"({n_iter / avg:.1f} it/s), avg of {n_trials} trials [synchronised]")


if __name__ == "__main__":
    main()
