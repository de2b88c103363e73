"""
Closed-form, bit-reproducible inputs for the BASELINE-length pins (tests/golden/longrun_*.npz).

The generator (gen_longrun.py, build container, imports the reference) and the GPU tests (tests/test_longrun_pins.py,
GPU box, no reference) must feed the SAME bits to the reference and to the engine, at 12 MP, without shipping
148-MB arrays.  So every value here is made of exactly-rounded IEEE operations only: PCG64 uniforms, float32
+ - * /, max, comparisons.  No exp / pow / norm / FFT (their results may differ by an ulp between CPUs and library
builds, and an ulp of input is visible at the float64 pins' 1e-9).  ``fingerprint`` (CRC-32 + float64 sum) is stored in
each fixture and re-checked on the GPU box before anything is compared.

NumPy only; imports nothing from the repo, the oracle or the reference.
"""
import math
import zlib

import numpy as np


def psf12(d, h, w, c, seed):
    """Sparse caustic-like PSF (``rng.random(...)**12``), plane by plane.  The twelfth power
    is three squarings and a product; the L2 normalisation of lensless/utils/io.py:375 becomes the constant
    1/sqrt(E[sum u^24]) = sqrt(25 / (h w c)) (a unit-energy PSF to 1e-3, which is all the solvers care about)."""
    rng = np.random.default_rng(seed)
    x = rng.random((d, h, w, c), dtype=np.float32)
    x2 = x * x
    x4 = x2 * x2
    x8 = x4 * x4
    scale = np.float32(math.sqrt(25.0 / (h * w * c)))
    return (x8 * x4) * scale


def bumps(h, w, c, seed, n, rmin, rmax):
    """n compact bumps max(0, 1 - r^2/R^2)^2 with random centres inside the central 60 % of the frame, radii
    R in [rmin, rmax] * min(h, w) and random per-channel weights.  float32, exact operations, fixed order."""
    rng = np.random.default_rng(seed)
    yy = np.arange(h, dtype=np.float32)[:, None]
    xx = np.arange(w, dtype=np.float32)[None, :]
    out = np.zeros((h, w, c), dtype=np.float32)
    for _ in range(n):
        cy = np.float32((0.2 + 0.6 * rng.random()) * h)
        cx = np.float32((0.2 + 0.6 * rng.random()) * w)
        r = (rmin + (rmax - rmin) * rng.random()) * min(h, w)
        inv_r2 = np.float32(1.0 / (r * r))
        wgt = rng.random(c, dtype=np.float32)
        dy = yy - cy
        dx = xx - cx
        b = np.maximum(np.float32(0), np.float32(1) - (dy * dy + dx * dx) * inv_r2)
        b = b * b
        out += b[:, :, None] * wgt[None, None, :]
    return out


def scene(h, w, c, seed=1):
    """What PSNR is quoted against (lensless/eval/metric.py:147-172 normalises both images by their own max)."""
    return bumps(h, w, c, seed, 12, 0.02, 0.07)


def measurement(h, w, c, seed):
    """A lensless-measurement-like frame: broad overlapping bumps (a scene seen through a caustic PSF is a smooth
    superposition) + uniform sensor noise, clipped at 0 and divided by its maximum (lensless/utils/io.py:196-197)."""
    base = bumps(h, w, c, 7000 + seed, 9, 0.25, 0.6)
    noise = np.random.default_rng(9000 + seed).random((h, w, c), dtype=np.float32)
    y = base + np.float32(0.02) * noise
    y = np.maximum(y, np.float32(0))
    return y / y.max()


def fingerprint(a):
    a = np.ascontiguousarray(a)
    return np.array([float(zlib.crc32(a.tobytes())), float(a.astype(np.float64).sum())])


def sample_points(h, w, n=8, size=32):
    """Top-left corners of the n crops: the four corners of the frame, its centre, and fixed interior points."""
    pts = [(0, 0), (0, w - size), (h - size, 0), (h - size, w - size), ((h - size) // 2, (w - size) // 2)]
    rng = np.random.default_rng(4242)
    while len(pts) < n:
        pts.append((int(rng.integers(0, h - size)), int(rng.integers(0, w - size))))
    return pts[:n]


def samples(img, size=32, stride=61):
    """(crops, lattice) of an (H, W, C) image: n size x size crops + every stride-th pixel of the whole frame."""
    h, w = img.shape[0], img.shape[1]
    crops = np.stack([np.asarray(img[y:y + size, x:x + size]) for y, x in sample_points(h, w, size=size)])
    return crops, np.asarray(img[::stride, ::stride])


def stats(img, ref_scene):
    """[sum, sum of squares, max, min, PSNR vs the scene] in float64 (PSNR: both divided by their own max)."""
    a = np.asarray(img, dtype=np.float64)
    s = np.asarray(ref_scene, dtype=np.float64)
    mse = np.mean((a / a.max() - s / s.max()) ** 2)
    return np.array([a.sum(), (a * a).sum(), a.max(), a.min(), 10.0 * math.log10(1.0 / mse)])
