"""
ctypes binding of the C ABI declared in ``include/lpc.h``.

The product libraries are ``lenslesspicam_amd/_lib/liblpc.so`` (float32) and ``liblpc_f64.so``
(float64; the same sources with -DLPC_DOUBLE), built by hipcc for gfx950.  There is no
CPU fallback: if the library is missing, or no HIP device is visible, the package fails
loudly.  ``Lib`` takes an explicit path so that the test-suite can drive *other builds of
the same C ABI* (the SIMT-emulator build under ``tests/simt_emu``) through the identical
binding; nothing in this package ever does that.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "_lib", "liblpc.so")          # float32 build
DEFAULT_LIB_F64 = os.path.join(_HERE, "_lib", "liblpc_f64.so")  # same sources, -DLPC_DOUBLE

ALGO_CONV, ALGO_ADMM, ALGO_GD, ALGO_NESTEROV, ALGO_FISTA = range(5)
NORM = {"backward": 0, "ortho": 1, "forward": 2}
K_SPATIAL, K_ROW_FWD, K_COL_A_FWD, K_COL_MID, K_COL_A_INV, K_ROW_INV, K_COUNT = range(7)
KERNEL_NAMES = ["spatial", "row_fwd", "col_a_fwd", "col_mid", "col_a_inv", "row_inv"]


# Launch-plan options every new handle starts from (include/lpc.h, lpc_config.options); an explicit ``options=`` /
# ``engine_options=`` argument overrides entry by entry.  Empty in production; the test-suite patches it to select a
# launch plan for solvers that shared helpers construct.
DEFAULT_OPTIONS: dict = {}


def _options_dict(opts) -> dict:
    if not opts:
        return {}
    if isinstance(opts, dict):
        return dict(opts)
    out = {}
    for tok in str(opts).replace(";", ",").replace(" ", ",").split(","):
        if tok:
            k, _, v = tok.partition("=")
            out[k] = v if v != "" else 1
    return out


PATH_OPTIONS = ("module_dir", "compiler")


def _option_value(key, v) -> str:
    """one value of the option string; the separators of the option syntax inside a path travel as %XX escapes
    (csrc/lpc_plan.h: unescape_opt_path)"""
    if isinstance(v, bool):
        return str(int(v))
    v = str(v)
    if key in PATH_OPTIONS:
        return "".join(f"%{ord(ch):02X}" if ch in "%,; \t\n" else ch for ch in v)
    if any(ch in v for ch in ",; \t\n"):
        raise ValueError(f"engine option {key}={v!r}: separators are only allowed in path-valued options")
    return v


class Config(C.Structure):
    _fields_ = [
        ("algo", C.c_int),
        ("height", C.c_int),
        ("width", C.c_int),
        ("channels", C.c_int),
        ("depth", C.c_int),
        ("batch", C.c_int),
        ("norm", C.c_int),
        ("pad", C.c_int),
        ("mu1", C.c_double),
        ("mu2", C.c_double),
        ("mu3", C.c_double),
        ("tau", C.c_double),
        ("lip_fact", C.c_double),
        ("nesterov_mu", C.c_double),
        ("nesterov_p", C.c_double),
        ("fista_tk", C.c_double),
        ("options", C.c_char_p),
    ]


class PrepConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("raw_type", "height", "width", "channels", "flip_ud", "flip_lr", "bgr_input",
                                       "gray", "normalize", "single_psf", "out_channels", "bg_pix0", "bg_pix1")]


RAW_TYPES = {"uint8": 0, "uint16": 1, "float32": 2, "float64": 3}


class NativeError(RuntimeError):
    pass


class Lib:
    """One loaded build of the C ABI."""

    def __init__(self, path: str = DEFAULT_LIB):
        if not os.path.exists(path):
            raise NativeError(
                f"native library not found: {path}. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)."
            )
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.c_void_p
        sig = {
            "lpc_create": [C.POINTER(Config), C.POINTER(vp)],
            "lpc_destroy": [vp],
            "lpc_plan_module": [C.POINTER(Config), C.c_int, C.c_char_p, C.c_size_t],
            "lpc_padded_shape": [vp, ip, ip, ip, ip],
            "lpc_set_psf": [vp, fp, vp],
            "lpc_convolve": [vp, fp, fp, C.c_int, C.c_int, C.c_int, vp],
            "lpc_convolve_spectrum": [vp, fp, fp, C.c_int, C.c_int, C.c_int, vp],
            "lpc_set_data": [vp, fp, C.c_int, vp],
            "lpc_set_initial_estimate": [vp, fp, vp],
            "lpc_reset": [vp, vp],
            "lpc_set_momentum": [vp, C.c_double, C.c_double, C.c_double],
            "lpc_iterate": [vp, C.c_int, vp],
            "lpc_admm_pnp_begin": [vp, C.c_int, fp, vp],
            "lpc_admm_pnp_end": [vp, C.c_int, fp, vp],
            "lpc_set_psi_gram": [vp, fp, vp],
            "lpc_admm_psi_step": [vp, fp, vp],
            "lpc_iterate_begin": [vp, vp],
            "lpc_iterate_end": [vp, fp, vp],
            "lpc_set_admm_schedule": [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double)],
            "lpc_set_fista_schedule": [vp, C.c_int, vp, vp, vp],
            "lpc_form_image": [vp, fp, vp],
            "lpc_get_state": [vp, C.c_char_p, fp, vp],
            "lpc_profile_enable": [vp, C.c_int],
            "lpc_profile_read": [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)],
            "lpc_kernel_bytes": [vp, C.c_int, C.POINTER(C.c_double)],
            "lpc_workspace_bytes": [vp, C.POINTER(C.c_size_t)],
            "lpc_model_bytes": [vp, C.POINTER(C.c_double)],
            "lpc_plan_info": [vp, C.c_char_p, C.c_size_t],
            "lpc_reconstruction_error": [vp, fp, fp, C.c_int, fp, vp],
            "lpc_image_metrics": [fp, fp, C.c_long, C.c_int, C.c_int, fp, vp],
            "lpc_preprocess_frames": [C.POINTER(PrepConfig), vp, C.c_int, fp, fp, vp],
            "lpc_preprocess_psf": [C.POINTER(PrepConfig), vp, C.c_int, fp, fp, vp],
            "lpc_resize_aa": [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, vp],
        }
        for name, args in sig.items():
            fn = getattr(d, name)
            fn.argtypes = args
            fn.restype = C.c_int
        d.lpc_last_error.restype = C.c_char_p
        d.lpc_backend.restype = C.c_char_p
        d.lpc_real_name.restype = C.c_char_p
        self.real = d.lpc_real_name().decode()            # "float32" | "float64"
        self.c_real = C.c_double if self.real == "float64" else C.c_float

    # -- plumbing -------------------------------------------------------------------
    def backend(self) -> str:
        return self.dll.lpc_backend().decode()

    def check(self, rc: int):
        if rc != 0:
            raise NativeError(self.dll.lpc_last_error().decode())

    def image_metrics(self, true_ptr, est_ptr, n, n_items, normalize, out_ptr, stream=0):
        self.check(self.dll.lpc_image_metrics(true_ptr, est_ptr, int(n), int(n_items), int(normalize), out_ptr,
                                              stream))

    def preprocess_frames(self, cfg: PrepConfig, raw_ptr, n, bg_ptr, out_ptr, stream=0):
        self.check(self.dll.lpc_preprocess_frames(C.byref(cfg), raw_ptr, int(n), bg_ptr, out_ptr, stream))

    def preprocess_psf(self, cfg: PrepConfig, raw_ptr, depth, psf_ptr, bg_ptr, stream=0):
        self.check(self.dll.lpc_preprocess_psf(C.byref(cfg), raw_ptr, int(depth), psf_ptr, bg_ptr, stream))

    def resize_aa(self, in_ptr, n, H, W, Cn, Hout, Wout, out_ptr, stream=0):
        self.check(self.dll.lpc_resize_aa(in_ptr, int(n), int(H), int(W), int(Cn), int(Hout), int(Wout), out_ptr, stream))

    @staticmethod
    def _config(kw) -> Config:
        cfg = Config()
        defaults = dict(algo=ALGO_ADMM, height=0, width=0, channels=3, depth=1, batch=1, norm=0, pad=1,
                        mu1=1e-6, mu2=1e-5, mu3=4e-5, tau=1e-4, lip_fact=1.8, nesterov_mu=0.9,
                        nesterov_p=0.0, fista_tk=1.0, options=None)
        defaults.update(kw)
        opts = {**DEFAULT_OPTIONS, **_options_dict(defaults["options"])}    # {"hv_full": 1} -> "hv_full=1"
        opts = ",".join(f"{k}={_option_value(k, v)}" for k, v in opts.items())
        defaults["options"] = opts.encode() if opts else None
        for k, v in defaults.items():
            setattr(cfg, k, v)
        return cfg

    def plan_module(self, build=False, **kw) -> str:
        """Key of the plan module ``create(**kw)`` would use ('' = run-time plans); ``build``: compile it if missing.
        Needs no device (lpc_plan_module)."""
        cfg = self._config(kw)
        buf = C.create_string_buffer(1024)
        self.check(self.dll.lpc_plan_module(C.byref(cfg), int(bool(build)), buf, len(buf)))
        return buf.value.decode()

    def create(self, **kw) -> "Handle":
        cfg = self._config(kw)
        h = C.c_void_p()
        self.check(self.dll.lpc_create(C.byref(cfg), C.byref(h)))
        handle = Handle(self, h, cfg)
        reason = handle.fallback_reason()
        if reason and reason not in _warned:           # a frame that SHOULD run on a plan module and does not
            _warned.add(reason)
            import warnings

            warnings.warn("lenslesspicam_amd: compile-time plans unavailable, this solver runs on the slower run-time "
                          f"plans ({reason[:300]})", RuntimeWarning, stacklevel=3)
        return handle


class Handle:
    """RAII wrapper around an ``lpc_handle``.  Pointers are raw device addresses (ints)."""

    def __init__(self, lib: Lib, h, cfg: Config):
        self.lib, self.h, self.cfg = lib, h, cfg
        hp, wp, sh, sw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.check(lib.dll.lpc_padded_shape(h, C.byref(hp), C.byref(wp), C.byref(sh), C.byref(sw)))
        self.Hp, self.Wp, self.sh, self.sw = hp.value, wp.value, sh.value, sw.value

    def close(self):
        if self.h is not None and self.h.value:
            self.lib.dll.lpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _c(self, rc):
        self.lib.check(rc)

    def set_psf(self, ptr, stream=0):
        self._c(self.lib.dll.lpc_set_psf(self.h, ptr, stream))

    def convolve(self, x_ptr, out_ptr, n, x_channels, adjoint, stream=0):
        self._c(self.lib.dll.lpc_convolve(self.h, x_ptr, out_ptr, n, int(x_channels), int(adjoint), stream))

    def convolve_spectrum(self, x_ptr, out_ptr, n, x_channels, adjoint, stream=0):
        self._c(self.lib.dll.lpc_convolve_spectrum(self.h, x_ptr, out_ptr, n, int(x_channels), int(adjoint), stream))

    def set_data(self, ptr, channels, stream=0):
        self._c(self.lib.dll.lpc_set_data(self.h, ptr, int(channels), stream))

    def set_initial_estimate(self, ptr, stream=0):
        self._c(self.lib.dll.lpc_set_initial_estimate(self.h, ptr, stream))

    def reset(self, stream=0):
        self._c(self.lib.dll.lpc_reset(self.h, stream))

    def set_momentum(self, p=0.0, mu=0.9, tk=0.0):
        self._c(self.lib.dll.lpc_set_momentum(self.h, p, mu, tk))

    def set_admm_schedule(self, mu1, mu2, mu3, tau):
        n = len(mu1)
        arr = [(C.c_double * n)(*[float(v) for v in a]) for a in (mu1, mu2, mu3, tau)]
        self._c(self.lib.dll.lpc_set_admm_schedule(self.h, n, *arr))

    def set_fista_schedule(self, alpha, coef, stream=0):
        """alpha: (n, C) floats, coef: (n,) floats (host)"""
        n = len(coef)
        flat = [float(v) for row in alpha for v in row]
        a = (self.lib.c_real * len(flat))(*flat)
        c = (self.lib.c_real * n)(*[float(v) for v in coef])
        self._c(self.lib.dll.lpc_set_fista_schedule(self.h, n, C.cast(a, C.c_void_p), C.cast(c, C.c_void_p), stream))

    def clear_admm_schedule(self):
        self._c(self.lib.dll.lpc_set_admm_schedule(self.h, 0, None, None, None, None))

    def iterate(self, n, stream=0):
        self._c(self.lib.dll.lpc_iterate(self.h, int(n), stream))

    def admm_pnp_begin(self, use_dual, out_ptr, stream=0):
        self._c(self.lib.dll.lpc_admm_pnp_begin(self.h, int(bool(use_dual)), out_ptr, stream))

    def admm_pnp_end(self, use_dual, u_ptr, stream=0):
        self._c(self.lib.dll.lpc_admm_pnp_end(self.h, int(bool(use_dual)), u_ptr, stream))

    def set_psi_gram(self, gabs_ptr, stream=0):
        self._c(self.lib.dll.lpc_set_psi_gram(self.h, gabs_ptr, stream))

    def admm_psi_step(self, psit_ptr, stream=0):
        self._c(self.lib.dll.lpc_admm_psi_step(self.h, psit_ptr, stream))

    def iterate_begin(self, stream=0):
        self._c(self.lib.dll.lpc_iterate_begin(self.h, stream))

    def iterate_end(self, proj_ptr, stream=0):
        self._c(self.lib.dll.lpc_iterate_end(self.h, proj_ptr, stream))

    def form_image(self, out_ptr, stream=0):
        self._c(self.lib.dll.lpc_form_image(self.h, out_ptr, stream))

    def get_state(self, name, out_ptr, stream=0):
        self._c(self.lib.dll.lpc_get_state(self.h, name.encode(), out_ptr, stream))

    def reconstruction_error(self, pred_ptr, data_ptr, normalize, out_ptr, stream=0):
        self._c(self.lib.dll.lpc_reconstruction_error(self.h, pred_ptr, data_ptr, int(normalize), out_ptr, stream))

    def profile_enable(self, on=True, kernels=None):
        """HIP events around the hot-loop launches: all of them, or only the kernel ids named in ``kernels``
        (names from KERNEL_NAMES) -- what a timed region can afford (include/lpc.h)."""
        flag = int(bool(on))
        if on and kernels is not None:
            flag = sum(2 << KERNEL_NAMES.index(k) for k in kernels)
        self._c(self.lib.dll.lpc_profile_enable(self.h, flag))

    def profile_read(self):
        ms = (C.c_double * K_COUNT)()
        n = (C.c_long * K_COUNT)()
        self._c(self.lib.dll.lpc_profile_read(self.h, ms, n))
        return {KERNEL_NAMES[k]: (ms[k], n[k]) for k in range(K_COUNT)}

    def kernel_bytes(self, kid):
        b = C.c_double()
        self._c(self.lib.dll.lpc_kernel_bytes(self.h, kid, C.byref(b)))
        return b.value

    def plan_info(self):
        buf = C.create_string_buffer(8192)       # (may carry a compiler log when a module failed to build)
        self._c(self.lib.dll.lpc_plan_info(self.h, buf, len(buf)))
        return buf.value.decode()

    def fallback_reason(self) -> str:
        """'' when the handle runs the plan it asked for; else why its plan module is missing (no compiler, a failed
        build, jit=0 ...).  Frames below jit_min_points and no_static=1 never ask for a module."""
        info = self.plan_info()
        if "; run-time plans (" not in info:
            return ""
        reason = info.split("; run-time plans (", 1)[1].rsplit(")", 1)[0]
        return "" if reason in ("no_static", "small frame") else reason

    def model_bytes(self):
        b = C.c_double()
        self._c(self.lib.dll.lpc_model_bytes(self.h, C.byref(b)))
        return b.value

    def workspace_bytes(self):
        b = C.c_size_t()
        self._c(self.lib.dll.lpc_workspace_bytes(self.h, C.byref(b)))
        return b.value


_default = {}
_warned = set()


def default_lib(dtype: str = "float32") -> Lib:
    """The product library (HIP) for ``dtype``.  Imports torch first so that the HIP runtime the
    extension binds to is the one torch already loaded (same soname), then refuses to run without a GPU."""
    if dtype not in _default:
        import torch

        if not torch.cuda.is_available():
            raise NativeError(
                "lenslesspicam_amd needs a HIP device (MI355X); none is visible and there is no CPU path."
            )
        torch.cuda.init()
        path = DEFAULT_LIB_F64 if dtype == "float64" else DEFAULT_LIB
        from . import build as _build

        if _build.is_stale():
            # fresh checkout on a GPU box (the .so is git-ignored), or sources edited since the last build: compile the
            # HIP sources in-tree before anything runs -- never a library that does not match the sources next to it.
            # This builds the product library itself; there still is no other execution path.
            _build.build_hip(force=True, verbose=False)
        lib = Lib(path)
        if not lib.backend().startswith("hip"):
            raise NativeError(f"refusing non-HIP backend {lib.backend()!r} in the product path")
        assert lib.real == dtype, (lib.real, dtype)
        _default[dtype] = lib
    return _default[dtype]
