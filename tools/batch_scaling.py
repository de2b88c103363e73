"""Frame-iterations per second of batched ADMM against the batch size (does a small batch live in the 256 MiB Infinity
Cache?): PYTHONPATH=. python tools/batch_scaling.py [H W]"""
import sys
import time

import numpy as np
import torch

import lenslesspicam_amd as lpa


def synthetic_psf(H, W, C, seed=1):
    """caustic-like random pattern, unit l2 norm (timing input only)"""
    p = np.random.default_rng(seed).random((1, H, W, C), dtype=np.float32) ** 8
    return p / np.linalg.norm(p.ravel())


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (270, 480)
    psf = torch.from_numpy(synthetic_psf(H, W, 3)).cuda()
    for B in (1, 2, 4, 8, 16, 32, 64, 128):
        rec = lpa.ADMM(psf)
        rec.set_data(torch.rand((B, 1, H, W, 3), device="cuda"))
        rec.apply_batch(n_iter=20)
        torch.cuda.synchronize()
        reps = max(2, 128 // B)
        t0 = time.perf_counter()
        for _ in range(reps):
            rec.apply_batch(n_iter=20)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"B={B:4d}: {dt * 1e3:8.3f} ms per 20 it   {B * 20 / dt:10.1f} frame-it/s", flush=True)
        del rec


if __name__ == "__main__":
    main()
