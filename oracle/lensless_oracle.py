"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A from-scratch CPU restatement (torch-CPU, float32 or float64) of the one hot
path of LCAV/LenslessPiCam that this repository accelerates: FFT convolution
with a fixed PSF, ADMM with anisotropic TV + non-negativity, and the projected
gradient-descent family (vanilla / Nesterov / FISTA).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the *checker* (or the timed CPU
baseline) -- never as the thing that is shipped.  The product package
``lenslesspicam_amd`` must not import it (tests/test_abi_and_layout.py enforces that).

Parity pin: the reference's own tests hold no numerical values for this path
(SURVEY.md section 8c), so this oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/gen_golden.py``
(which imports /root/reference read-only) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it restates (paths relative to the
reference checkout).  The operation ORDER of the float arithmetic follows the
reference so that float32 trajectories agree to round-off of the FFT library.
"""

from __future__ import annotations

import math

import numpy as np
import torch

# --------------------------------------------------------------------------
# geometry: lensless/recon/rfft_convolve.py:102-117
# --------------------------------------------------------------------------


def next_fast_len_5smooth(n: int) -> int:
    """Smallest m >= n of the form 2^a 3^b 5^c.

    Restates what ``scipy.fftpack.next_fast_len`` returns (rfft_convolve.py:17,112
    uses the *fftpack* flavour, which is 5-smooth, not pocketfft's 11-smooth one).
    """
    m = max(int(n), 1)
    while True:
        r = m
        for p in (2, 3, 5):
            while r % p == 0:
                r //= p
        if r == 1:
            return m
        m += 1


class Geometry:
    """Padded FFT geometry for a PSF of spatial size (h, w).

    rfft_convolve.py:110-117: padded = next_fast_len(2*dim - 1) per axis,
    start = (padded - dim) // 2, end = start + dim.
    """

    def __init__(self, h: int, w: int):
        self.h, self.w = int(h), int(w)
        self.hp = next_fast_len_5smooth(2 * self.h - 1)
        self.wp = next_fast_len_5smooth(2 * self.w - 1)
        self.sh = (self.hp - self.h) // 2
        self.sw = (self.wp - self.w) // 2
        self.eh = self.sh + self.h
        self.ew = self.sw + self.w

    def pad(self, v: torch.Tensor) -> torch.Tensor:
        """rfft_convolve.py:84-100 (zero-embed, channels last)."""
        shape = list(v.shape)
        shape[-3], shape[-2] = self.hp, self.wp
        out = torch.zeros(shape, dtype=v.dtype)
        out[..., self.sh : self.eh, self.sw : self.ew, :] = v
        return out

    def crop(self, x: torch.Tensor) -> torch.Tensor:
        """rfft_convolve.py:79-82 (a view, like the reference)."""
        return x[..., self.sh : self.eh, self.sw : self.ew, :]


def _as_tensor(a, dtype):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return a.to(dtype=dtype, device="cpu")


# --------------------------------------------------------------------------
# linear operator: lensless/recon/rfft_convolve.py:27-223
# --------------------------------------------------------------------------


class ConvolverOracle:
    """FFT convolution with a fixed real PSF of shape (D, H, W, C)."""

    def __init__(self, psf, dtype=torch.float32, pad=True, norm="ortho"):
        psf = _as_tensor(psf, dtype)
        assert psf.dim() == 4
        self.dtype = dtype
        self.pad_flag = pad
        self.norm = norm
        self.psf = psf
        self.geom = Geometry(psf.shape[-3], psf.shape[-2])
        g = self.geom
        # rfft_convolve.py:121 -- spectrum of the zero-embedded PSF, halved along W
        self.H = torch.fft.rfft2(g.pad(psf), norm=norm, dim=(-3, -2), s=(g.hp, g.wp))
        self.Hadj = torch.conj(self.H)

    def _apply(self, x, spec):
        g = self.geom
        xp = g.pad(x) if self.pad_flag else x
        # rfft_convolve.py:145-155 / 190-200: un-normalised forward, 1/N inverse, then ifftshift
        y = torch.fft.rfft2(xp, dim=(-3, -2)) * spec
        y = torch.fft.ifftshift(torch.fft.irfft2(y, dim=(-3, -2), s=(g.hp, g.wp)), dim=(-3, -2))
        return g.crop(y) if self.pad_flag else y

    def convolve(self, x):
        return self._apply(x, self.H)

    def deconvolve(self, y):
        return self._apply(y, self.Hadj)


# --------------------------------------------------------------------------
# TV helpers: lensless/recon/admm.py:341-397
# --------------------------------------------------------------------------


def soft_thresh(x, thresh):
    """admm.py:341-346."""
    return torch.sign(x) * torch.max(torch.abs(x) - thresh, torch.zeros_like(x))


def finite_diff(x):
    """admm.py:349-359: circular backward differences along rows (-3) and cols (-2)."""
    return torch.stack(
        (torch.roll(x, 1, dims=-3) - x, torch.roll(x, 1, dims=-2) - x), dim=x.dim()
    )


def finite_diff_adj(u):
    """admm.py:362-370."""
    d1 = torch.roll(u[..., 0], -1, dims=-3) - u[..., 0]
    d2 = torch.roll(u[..., 1], -1, dims=-2) - u[..., 1]
    return d1 + d2


def finite_diff_gram(padded_shape, dtype):
    """admm.py:373-397 for depth 1: rfft2 of the 5-point stencil, all channels."""
    gram = torch.zeros(list(padded_shape), dtype=dtype)
    assert padded_shape[0] == 1
    gram[0, 0, 0] = 4
    gram[0, 0, 1] = gram[0, 0, -1] = gram[0, 1, 0] = gram[0, -1, 0] = -1
    return torch.fft.rfft2(gram, dim=(-3, -2))


# --------------------------------------------------------------------------
# ADMM: lensless/recon/admm.py:35-338 driven by recon.py:498-605
# --------------------------------------------------------------------------


class ADMMOracle:
    """ADMM (TV prior + non-negativity) for a depth-1 PSF.

    State names follow the reference attributes (V = ``_image_est`` etc.).
    """

    def __init__(self, psf, dtype=torch.float32, mu1=1e-6, mu2=1e-5, mu3=4e-5, tau=1e-4,
                 initial_est=None, schedule=None, denoiser=None, psi=None):
        """``denoiser``: optional (fn, noise_level, use_dual) -- the plug-and-play branch of admm.py:126-133,
        235-243,266-275,300-311, restated as written (U and eta image-shaped, Psi^T = identity).
        ``schedule``: optional dict of per-iteration sequences mu1/mu2/mu3/tau -- the arithmetic of
        ``lensless/recon/unrolled_admm.py:133-234`` (UnrolledADMM inference, no pre/post processors):
        iteration i uses the i-th entries everywhere, R_divmat and X_divmat included."""
        """``psi``: optional (psi, psi_adj, psi_gram) callables, the caller-supplied prior of admm.py:44-46,104-120
        (``psi_gram(padded_shape)`` returns the rfft2 spectrum of Psi^T Psi)."""
        psf = _as_tensor(psf, dtype)
        assert psf.dim() == 4 and psf.shape[0] == 1, "reference refuses D>1 (admm.py:92-96)"
        self.dtype = dtype
        self.cdtype = torch.complex64 if dtype == torch.float32 else torch.complex128
        self.mu1, self.mu2, self.mu3, self.tau = mu1, mu2, mu3, tau
        # admm.py:101: pad=False, norm="backward"
        self.conv = ConvolverOracle(psf, dtype=dtype, pad=False, norm="backward")
        g = self.conv.geom
        self.geom = g
        self.padded_shape = [1, g.hp, g.wp, psf.shape[-1]]
        self.psf = psf
        self.Psi, self.PsiT = finite_diff, finite_diff_adj
        self.gram = finite_diff_gram(self.padded_shape, dtype)  # admm.py:107
        if psi is not None:                                     # admm.py:108-118
            self.Psi, self.PsiT = psi[0], psi[1]
            self.gram = psi[2](self.padded_shape)
        self.initial_est = None if initial_est is None else _as_tensor(initial_est, dtype)
        self.schedule = schedule
        self.denoiser = denoiser
        self.data = None
        self.reset()

    def set_data(self, data):
        data = _as_tensor(data, self.dtype)
        while data.dim() < 5:  # recon.py:376-381
            data = data[None]
        self.data = data

    def _set_params(self, it):
        """unrolled_admm.py:147-168: per-iteration parameters as float32 tensors (abs of the learnt values)."""
        if self.schedule is None:
            return
        t = lambda k: torch.abs(torch.tensor(self.schedule[k][it], dtype=self.dtype))  # noqa: E731
        self.mu1, self.mu2, self.mu3, self.tau = t("mu1"), t("mu2"), t("mu3"), t("tau")
        H, Hadj = self.conv.H, self.conv.Hadj
        self.R_divmat = 1.0 / (
            self.mu1 * torch.abs(Hadj * H) + self.mu2 * torch.abs(self.gram) + self.mu3
        ).type(self.cdtype)
        self.X_divmat = 1.0 / (self.geom.pad(torch.ones_like(self.psf)) + self.mu1)

    def reset(self):
        """admm.py:150-195."""
        self.it = 0
        if self.initial_est is not None:
            V = self.initial_est
            if V.dim() == 4:
                V = V[None]
            self.V = V
        else:
            self.V = torch.zeros([1] + self.padded_shape, dtype=self.dtype)
        self.X = torch.zeros_like(self.V)
        self.U = torch.zeros_like(self.V if self.denoiser is not None else self.Psi(self.V))  # admm.py:163-169
        self.W = torch.zeros_like(self.X)
        if self.V.max():  # admm.py:172
            self.HV = self.conv.convolve(self.V)
            self.PsiV = self.Psi(self.V)
        else:
            self.HV = torch.zeros_like(self.X)
            self.PsiV = torch.zeros_like(self.U)
        self.xi = torch.zeros_like(self.V)
        self.eta = torch.zeros_like(self.U)
        self.rho = torch.zeros_like(self.X)
        H, Hadj = self.conv.H, self.conv.Hadj
        # admm.py:186-190 (real values stored as complex)
        self.R_divmat = 1.0 / (
            self.mu1 * torch.abs(Hadj * H) + self.mu2 * torch.abs(self.gram) + self.mu3
        ).type(self.cdtype)
        # admm.py:193
        self.X_divmat = 1.0 / (self.geom.pad(torch.ones_like(self.psf)) + self.mu1)

    def step(self):
        """One iteration, admm.py:313-329 with the sub-updates at :232-311."""
        g = self.geom
        self._set_params(self.it)
        self.it += 1
        mu1, mu2, mu3 = self.mu1, self.mu2, self.mu3
        if self.denoiser is not None:
            return self._step_pnp()
        self.U = soft_thresh(self.PsiV + self.eta / mu2, self.tau / mu2)          # :245-247
        self.X = self.X_divmat * (self.xi + mu1 * self.HV + g.pad(self.data))     # :252-254
        self.W = torch.maximum(self.rho / mu3 + self.V, torch.zeros_like(self.V))  # :259-261
        rk = (
            (mu3 * self.W - self.rho)
            + self.PsiT(mu2 * self.U - self.eta)
            + self.conv.deconvolve(mu1 * self.X - self.xi)
        )                                                                         # :277-281
        freq = self.R_divmat * torch.fft.rfft2(rk, dim=(-3, -2))                  # :286
        self.V = torch.fft.irfft2(freq, dim=(-3, -2), s=(g.hp, g.wp))             # :287-289
        self.HV = self.conv.convolve(self.V)                                      # :320
        self.PsiV = self.Psi(self.V)                                              # :322
        self.xi = self.xi + mu1 * (self.HV - self.X)                              # :298-300
        self.eta = self.eta + mu2 * (self.PsiV - self.U)                          # :302-308
        self.rho = self.rho + mu3 * (self.V - self.W)                             # :310-311

    def _step_pnp(self):
        """The denoiser branch.  The conditional expression of admm.py:266-275 binds as
        ``(A + B - C) if use_dual else (D + E)``: the dual form has no data term, the other one no W term."""
        g = self.geom
        fn, nl, dual = self.denoiser
        mu1, mu2, mu3 = self.mu1, self.mu2, self.mu3
        self.U = fn(self.U + self.eta / mu2, nl) if dual else fn(self.V, nl)       # :235-243
        self.X = self.X_divmat * (self.xi + mu1 * self.HV + g.pad(self.data))     # :252-254
        self.W = torch.maximum(self.rho / mu3 + self.V, torch.zeros_like(self.V))  # :259-261
        if dual:
            rk = (mu3 * self.W - self.rho) + mu2 * self.U - self.eta
        else:
            rk = mu2 * self.U + self.conv.deconvolve(mu1 * self.X - self.xi)
        freq = self.R_divmat * torch.fft.rfft2(rk, dim=(-3, -2))
        self.V = torch.fft.irfft2(freq, dim=(-3, -2), s=(g.hp, g.wp))
        self.HV = self.conv.convolve(self.V)                                      # :320
        self.xi = self.xi + mu1 * (self.HV - self.X)
        if dual:
            self.eta = self.eta + mu2 * (self.V - self.U)                         # :304-306, only with use_dual (:326)
        self.rho = self.rho + mu3 * (self.V - self.W)

    def form_image(self):
        """admm.py:331-338: crop is a view, clamp happens IN PLACE on V.  (UnrolledADMM clips out of
        place instead, unrolled_admm.py:236-240.)"""
        img = self.geom.crop(self.V)
        if self.schedule is not None:
            return torch.clip(img, min=0.0)
        img[img < 0] = 0
        return img

    def apply(self, n_iter, reset=True, background=None, show=False, disp_iter=-1, ax=None):
        """recon.py:547-594: exactly n_iter updates, returns (D,H,W,C).  ``show`` stands for ``plot or save``: the
        image is then formed before the loop (when no ``ax`` is handed in) and after every iteration i with
        ``(i + 1) % disp_iter == 0`` (recon.py:563-584; Python's modulo: disp_iter=-1 is every iteration) -- each of
        those read-outs clamps V in place, and right after ``reset()`` V still aliases the initial estimate."""
        assert self.data is not None and self.data.shape[0] == 1
        if background is not None:  # recon.py:553-555
            self.data = self.data - _as_tensor(background, self.dtype)
            self.data[self.data < 0] = 0
        if reset:
            self.reset()
        if show and disp_iter is not None:          # recon.py:563-566
            if ax is None:
                self.form_image()
        else:                                       # recon.py:568-570
            disp_iter = n_iter + 1
        for i in range(n_iter):
            self.step()
            if show and (i + 1) % disp_iter == 0:   # recon.py:580-582
                self.form_image()
        return self.form_image()[0]


# --------------------------------------------------------------------------
# gradient descent family: lensless/recon/gd.py:62-241
# --------------------------------------------------------------------------


class GDOracle:
    """kind in {"vanilla", "nesterov", "fista"}; PSF may have depth D >= 1."""

    def __init__(self, psf, kind="fista", dtype=torch.float32, lip_fact=1.8, mu=0.9, p=0.0,
                 tk=1.0, initial_est=None, norm="ortho", proj=None):
        """proj: the projection applied by ``_form_image`` (gd.py:136-140); None = non_neg (gd.py:41-59).  An
        external denoiser is the same thing with its noise level bound (gd.py:89-92)."""
        assert kind in ("vanilla", "nesterov", "fista")
        self.proj = proj if proj is not None else (lambda x: torch.maximum(x, torch.zeros_like(x)))
        psf = _as_tensor(psf, dtype)
        assert psf.dim() == 4
        self.kind, self.dtype = kind, dtype
        self.lip_fact, self.mu, self.p0, self.tk0 = lip_fact, mu, p, tk
        self.psf = psf
        self.conv = ConvolverOracle(psf, dtype=dtype, pad=True, norm=norm)  # recon.py:207,293
        self.geom = self.conv.geom
        self.initial_est = None if initial_est is None else _as_tensor(initial_est, dtype)
        self.data = None
        self.reset()

    def set_data(self, data):
        data = _as_tensor(data, self.dtype)
        while data.dim() < 5:
            data = data[None]
        self.data = data

    def reset(self, p=0.0, mu=0.9, tk=None):
        """gd.py:94-126 plus the subclass resets.

        Quirk restated on purpose: ``NesterovGradientDescent.reset(p=0, mu=0.9)``
        (gd.py:178-181) overwrites whatever ``mu``/``p`` the constructor received,
        because the base constructor calls ``reset()`` with no arguments
        (recon.py:328-329).  ``FISTA.reset(tk=None)`` falls back to the constructor's
        ``tk`` (gd.py:227-232).
        """
        C = self.psf.shape[3]
        if self.initial_est is not None:
            x = self.initial_est
            if x.dim() == 4:
                x = x[None]
            self.x = x.clone()
        else:
            flat = self.psf.reshape(-1, C)                                   # gd.py:100-105
            start = (torch.max(flat, dim=0).values + torch.min(flat, dim=0).values) / 2
            self.x = torch.ones_like(self.psf[None]) * start
        Hf = self.conv.H.reshape(-1, C)
        Haf = self.conv.Hadj.reshape(-1, C)
        # gd.py:107-112: alpha = lip_fact / max |H* H| per channel
        self.alpha = torch.real(self.lip_fact / torch.max(torch.abs(Haf * Hf), dim=0).values)
        self.p = p                                                           # gd.py:179-181
        self.mu = mu
        self.tk = tk if tk else self.tk0                                     # gd.py:227-233
        # gd.py:233: ``_xk`` is bound to the SAME tensor as ``_image_est``; the first
        # in-place ``-=`` (gd.py:236) therefore also moves ``_xk``.  Harmless for the
        # default tk=1 ((tk-1)=0) but visible for tk != 1, so it is restated here.
        self.xk = None

    def grad(self):
        diff = self.conv.convolve(self.x) - self.data                        # gd.py:128-130
        return self.conv.deconvolve(diff)

    def step(self):
        if self.kind == "vanilla":                                           # gd.py:132-134
            self.x = self.x - self.alpha * self.grad()
            self.x = self.proj(self.x)
        elif self.kind == "nesterov":                                        # gd.py:183-188
            p_prev = self.p
            self.p = self.mu * self.p - self.alpha * self.grad()
            self.x = self.x + (-self.mu * p_prev + (1 + self.mu) * self.p)
            self.x = self.proj(self.x)
        else:                                                                # gd.py:235-241
            self.x = self.x - self.alpha * self.grad()
            xk = self.proj(self.x)
            tk = (1 + math.sqrt(1 + 4 * self.tk ** 2)) / 2
            xk_prev = self.x if self.xk is None else self.xk   # aliasing quirk, see reset()
            self.x = xk + (self.tk - 1) / tk * (xk - xk_prev)
            self.tk = tk
            self.xk = xk

    def form_image(self):
        return self.proj(self.x)                                             # gd.py:136-140,41-59

    def apply(self, n_iter, reset=True, background=None):
        assert self.data is not None and self.data.shape[0] == 1
        if background is not None:
            self.data = self.data - _as_tensor(background, self.dtype)
            self.data[self.data < 0] = 0
        if reset:
            self.reset()
        for _ in range(n_iter):
            self.step()
        return self.form_image()[0]


def unrolled_fista_oracle(psf, data, alpha, tk, dtype=torch.float32):
    """UnrolledFISTA inference, lensless/recon/unrolled_fista.py:60-106 (no pre/post processors).

    alpha: (n_iter, C) per-iteration steps, tk: (n_iter + 1,) momentum sequence; both enter through
    torch.abs() as float32 (:98-100).  x_k starts as the initial image (:91-96, no aliasing here because
    the update is out of place).  data: (H,W,C) or (1,H,W,C); returns (1,D,H,W,C).
    """
    g = GDOracle(psf, kind="fista", dtype=dtype)
    g.set_data(data)
    alpha = torch.abs(torch.as_tensor(np.asarray(alpha), dtype=dtype))
    tk = torch.abs(torch.as_tensor(np.asarray(tk), dtype=dtype))
    x = g.x
    xk_prev = x
    for i in range(alpha.shape[0]):
        g.x = x
        x = x - alpha[i] * g.grad()                                     # :103
        xk = torch.maximum(x, torch.zeros_like(x))                      # proj = non_neg
        x = xk + (tk[i] - 1) / tk[i + 1] * (xk - xk_prev)               # :105
        xk_prev = xk
    return torch.maximum(x, torch.zeros_like(x))                        # _form_image, :80-81


# --------------------------------------------------------------------------
# metrics used by the harness
# --------------------------------------------------------------------------


def reconstruction_error(conv: ConvolverOracle, prediction, lensless, normalize=True):
    """recon.py:607-653 restated.  conv: a pad=True ConvolverOracle; the reference's pad=False branch (ADMM)
    pads, convolves and crops, which is the same arithmetic.  prediction (B,D,H,W,C), lensless (B,1,H,W,C)."""
    Hx = conv.convolve(prediction)
    if normalize:
        Hx = Hx - torch.amin(Hx, dim=(-1, -2, -3), keepdim=True)
        Hx = Hx / torch.amax(Hx, dim=(-1, -2, -3), keepdim=True)
    npix = float(np.prod(conv.psf.shape))
    return torch.sum((Hx - lensless) ** 2, dim=(-1, -2, -3, -4)) / npix


def mse(true, est, normalize=True):
    """lensless/eval/metric.py:119-144: both images / their own max (float32), then skimage's
    mean_squared_error = float64 mean of the squared float32 difference."""
    a = np.array(true, dtype=np.float32)
    b = np.array(est, dtype=np.float32)
    if normalize:
        a = a / a.max()
        b = b / b.max()
    return float(np.mean((a - b) ** 2, dtype=np.float64))


def psnr_skimage(true, est, normalize=True):
    """lensless/eval/metric.py:147-172 with skimage.metrics.peak_signal_noise_ratio(data_range=None) restated:
    for float images the range is 1 when min(true) >= 0, else 2 (dtype range (-1, 1))."""
    a = np.array(true, dtype=np.float32)
    if normalize:
        a = a / a.max()
    r = 1.0 if a.min() >= 0 else 2.0
    return 10.0 * math.log10(r * r / mse(true, est, normalize))


def psnr(img, ref):
    """lensless/eval/metric.py:147-172 semantics: each image / its own max, 10 log10(1/MSE)."""
    a = np.asarray(img, dtype=np.float64)
    b = np.asarray(ref, dtype=np.float64)
    a = a / a.max()
    b = b / b.max()
    return 10.0 * math.log10(1.0 / float(np.mean((a - b) ** 2)))


# --------------------------------------------------------------------------
# synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------


def synthetic_psf(D, H, W, C, seed=0):
    """Sparse caustic-like PSF, L2-normalised like lensless/utils/io.py:375."""
    rng = np.random.default_rng(seed)
    psf = rng.random((D, H, W, C), dtype=np.float32) ** 12
    psf /= np.linalg.norm(psf.ravel())
    return psf.astype(np.float32)


def synthetic_scene(H, W, C, seed=1, nblobs=12):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    scene = np.zeros((H, W, C), dtype=np.float32)
    for _ in range(nblobs):
        cy = (0.2 + 0.6 * rng.random()) * H
        cx = (0.2 + 0.6 * rng.random()) * W
        s = (0.02 + 0.05 * rng.random()) * min(H, W)
        blob = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
        scene += blob[..., None] * rng.random(C, dtype=np.float32)[None, None, :]
    return scene


def synthetic_measurement(psf, scene):
    """raw = clip(crop(scene (*) psf), 0) / max  (io.py:196-197)."""
    conv = ConvolverOracle(psf[:1], pad=True, norm="backward")
    y = conv.convolve(torch.from_numpy(scene)[None, None])[0, 0].numpy()
    y = np.clip(y, 0, None)
    return (y / y.max()).astype(np.float32)
