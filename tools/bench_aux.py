#!/usr/bin/env python3
"""Timings of the rows added around the hot loop at the 12 MP frame:
raw-frame preparation, on-device evaluation reductions, and the per-iteration cost of the projection hook."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lenslesspicam_amd as lpa  # noqa: E402
from lenslesspicam_amd import metric, prep  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    H, W, C = 3040, 4056, 3
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    raw_psf = (torch.rand((1, H, W, C), generator=g) ** 8 * 3500 + 120).to(torch.int16).to(dev)
    raw_frm = (torch.rand((1, H, W, C), generator=g) * 3000 + 150).to(torch.int16).to(dev)
    ms = timed(lambda: prep.preprocess_data(raw_psf, raw_frm, flip=True, normalize=True))
    raw_b = 2 * raw_psf.numel() * 2
    out_b = 2 * raw_psf.numel() * 4
    print(f"preprocess_data 12MP uint16 RGB (PSF + frame): {ms:.3f} ms  "
          f"({(2 * raw_b + out_b) / ms / 1e6:.0f} GB/s over 2 raw reads + 1 float write)")
    psf, data = prep.preprocess_data(raw_psf, raw_frm, flip=True, normalize=True)
    rec = lpa.FISTA(psf)
    rec.set_data(data)
    rec.apply(n_iter=5, disp_iter=None)
    est = rec.get_image_estimate()
    ms = timed(lambda: rec.reconstruction_error(prediction=est))
    print(f"reconstruction_error 12MP (1 convolution + 3 reductions): {ms:.3f} ms")
    a = est[0]
    b = (a + 0.01 * torch.randn_like(a)).contiguous()
    ms = timed(lambda: metric.metrics_batch(a, b))
    print(f"mse+psnr 12MP pair: {ms:.3f} ms ({4 * a.numel() * 4 / ms / 1e6:.0f} GB/s over 2 passes of both images)")
    ms_f = timed(lambda: rec._iterate(10), n=3) / 10
    hook = lpa.FISTA(psf, proj=lambda x: torch.clamp(x, min=0.0))
    hook.set_data(data)
    ms_h = timed(lambda: hook._iterate(10), n=3) / 10
    print(f"FISTA iteration 12MP: fused {ms_f:.3f} ms, through the projection hook (torch.clamp) {ms_h:.3f} ms")


if __name__ == "__main__":
    main()
