"""How does the iteration rate scale with the number of colour planes resident per pass?  One plane of a 12-MP frame
(spectrum 201 MB) nearly fits the 256 MiB Infinity Cache, three do not: python tools/chan_scaling.py [H W]"""
import sys
import time

import numpy as np
import torch

import lenslesspicam_amd as lpa


def rate(cls, psf, y, n):
    rec = cls(psf)
    rec.set_data(y)
    rec.apply(n_iter=3, disp_iter=None, plot=False)
    torch.cuda.synchronize()
    rec.reset()
    t0 = time.perf_counter()
    rec.apply(n_iter=n, disp_iter=None, plot=False)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def synthetic_psf(H, W, C, seed=1):
    """caustic-like random pattern, unit l2 norm (timing input only)"""
    p = np.random.default_rng(seed).random((1, H, W, C), dtype=np.float32) ** 8
    return p / np.linalg.norm(p.ravel())


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3040, 4056)
    for C in (1, 3):
        psf = torch.from_numpy(synthetic_psf(H, W, C)).cuda()
        y = torch.rand((1, H, W, C), device="cuda")
        for name, cls, n in (("ADMM", lpa.ADMM, 40), ("FISTA", lpa.FISTA, 60)):
            r = rate(cls, psf, y, n)
            print(f"{H}x{W}x{C} {name}: {r:8.1f} it/s  = {r * C:8.1f} plane-it/s", flush=True)


if __name__ == "__main__":
    main()
