"""CPU-side checks: the product library loads and exports every symbol of include/lpc.h (no compute
without a GPU), the package fails loudly without a device, and the product package never touches the
oracle, the emulator or /root/reference."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "lpc.h")).read()
    return sorted(set(re.findall(r"\b(lpc_[a-z_]+)\s*\(", hdr)))


def test_hip_library_exports_every_header_symbol():
    from lenslesspicam_amd import build

    build.build_hip(force=False, verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    syms = _header_symbols()
    assert len(syms) >= 20
    for path, real in ((build.OUT, b"float32"), (build.OUT_F64, b"float64")):
        dll = ctypes.CDLL(path)
        for name in syms:
            assert hasattr(dll, name), (path, name)
        dll.lpc_backend.restype = ctypes.c_char_p
        dll.lpc_real_name.restype = ctypes.c_char_p
        assert dll.lpc_backend() == b"hip-gfx950" and dll.lpc_real_name() == real


def test_emulator_build_exports_the_same_abi(emu_lib):
    for name in _header_symbols():
        assert hasattr(emu_lib.dll, name), name
    assert "emu" in emu_lib.backend()


def test_norm_enumerators_match_the_binding_and_the_reference_strings():
    """enum lpc_norm (include/lpc.h) <-> _native.NORM <-> the three strings rfft_convolve.py:27,121 hands to rfft2"""
    from lenslesspicam_amd import _native

    hdr = open(os.path.join(ROOT, "include", "lpc.h")).read()
    enum = dict(re.findall(r"(LPC_NORM_[A-Z]+)\s*=\s*(\d+)", hdr))
    assert enum == {"LPC_NORM_BACKWARD": "0", "LPC_NORM_ORTHO": "1", "LPC_NORM_FORWARD": "2"}
    assert _native.NORM == {"backward": int(enum["LPC_NORM_BACKWARD"]), "ortho": int(enum["LPC_NORM_ORTHO"]),
                            "forward": int(enum["LPC_NORM_FORWARD"])}
    # every one of them is exercised against the reference: tests/test_norm_scale.py (ADMM, FISTA, operator)
    import test_norm_scale

    norms = {str(n) for si in (0, 1) for n in test_norm_scale.ladder()[f"s{si}_norms"]}
    assert norms == set(_native.NORM)


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_product_path_fails_loudly_without_gpu():
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd._native import NativeError

    with pytest.raises(NativeError, match="no CPU path"):
        lpa.ADMM(np.zeros((1, 8, 8, 3), np.float32))
    with pytest.raises(NativeError):
        lpa.RealFFTConvolve2D(np.zeros((1, 8, 8, 3), np.float32))


def test_product_package_never_references_checker_or_reference():
    pkg = os.path.join(ROOT, "lenslesspicam_amd")
    bad = re.compile(r"(^|\W)(import\s+oracle|from\s+oracle|simt_emu|/root/reference|liblpc_emu)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".inc")):
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            for ln in text.splitlines():
                if ln.strip().startswith(("//", "#", "*")) or "tests/simt_emu" in ln:
                    continue  # explanatory comments may mention the test harness
                assert not bad.search(ln), (f, ln)


def test_reference_api_surface_and_errors(backend):
    """constructor keywords/defaults, kwargs swallowing, error types."""
    import inspect

    import lenslesspicam_amd as lpa

    sig = inspect.signature(lpa.ADMM.__init__)
    assert [p for p in sig.parameters][1:12] == ["psf", "dtype", "mu1", "mu2", "mu3", "tau", "psi", "psi_adj",
                                                 "psi_gram", "pad", "norm"]
    assert (sig.parameters["mu1"].default, sig.parameters["mu2"].default, sig.parameters["mu3"].default,
            sig.parameters["tau"].default) == (1e-6, 1e-5, 4e-5, 1e-4)
    assert inspect.signature(lpa.FISTA.__init__).parameters["tk"].default == 1.0
    assert inspect.signature(lpa.NesterovGradientDescent.__init__).parameters["mu"].default == 0.9
    assert inspect.signature(lpa.GradientDescent.__init__).parameters["lip_fact"].default == 1.8
    ap = inspect.signature(lpa.ReconstructionAlgorithm.apply).parameters
    assert [p for p in ap][1:10] == ["n_iter", "disp_iter", "plot_pause", "plot", "save", "gamma", "ax", "reset",
                                      "background"]
    psf = np.random.default_rng(0).random((1, 10, 12, 3), dtype=np.float32)
    # scripts pass extra keys through **config.admm (configs/recon/defaults.yaml:60-82)
    rec = lpa.ADMM(psf, n_iter=3, unrolled=False, checkpoint_fp=None, pre_process_model=None, disp_iter=5)
    assert rec._n_iter == 3
    with pytest.raises(AssertionError):
        rec.apply()                                             # data not set
    with pytest.raises(AssertionError):
        rec.set_data(np.zeros((11, 12, 3), np.float32))         # shape mismatch
    with pytest.raises(AssertionError):
        rec.set_data(torch.zeros(10, 12, 3))                    # numpy PSF -> numpy data
    rec.set_data(np.zeros((2, 1, 10, 12, 3), np.float32))
    with pytest.raises(AssertionError):
        rec.apply(n_iter=1)                                     # apply() needs batch 1 (recon.py:549-551)
    with pytest.raises(AssertionError):
        lpa.ADMM(psf[0])                                        # PSF must be 4-D
    with pytest.raises(AssertionError):
        lpa.ADMM(np.zeros((1, 8, 8, 2), np.float32))            # C in {1,3}
    for dt in ("float32", "float64"):                           # test/test_algos.py:89-131 loops both dtypes
        r64 = lpa.GradientDescent(psf.astype(dt), dtype=dt, n_iter=2)
        r64.set_data(psf[0].astype(dt))
        assert r64.apply(disp_iter=None, plot=False).dtype == np.dtype(dt)
    with pytest.raises(AssertionError):
        lpa.ADMM(psf, dtype="float16")
    with pytest.raises(NotImplementedError):
        lpa.ADMM(psf, denoiser={"network": "DruNet", "noise_level": 10})
    with pytest.raises(AssertionError):                          # admm.py:109-110: psi needs psi_adj and psi_gram
        lpa.ADMM(psf, psi=lambda x: x)
    res = lpa.FISTA(psf, n_iter=2)
    res.set_data(psf[0])
    out = res.apply(disp_iter=None, plot=False)                 # n_iter from the constructor
    assert out.shape == (1, 10, 12, 3) and out.dtype == np.float32
    assert lpa.GradientDescentUpdate.all_values() == ["fista", "nesterov", "vanilla"]


@pytest.mark.gpu
def test_integration_md_stub_runs_verbatim():
    """INTEGRATION.md section B is documentation a maintainer would paste: execute its first code block as written
    (only the library path is made absolute) against liblpc.so and compare with the package's own ADMM."""
    import re

    import lenslesspicam_amd as lpa
    from lenslesspicam_amd import _native
    from oracle import lensless_oracle as orc

    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## B."):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'C.CDLL("liblpc.so")' in code
    _native.default_lib("float32")                       # builds the library on a fresh checkout
    code = code.replace('C.CDLL("liblpc.so")', f"C.CDLL({_native.DEFAULT_LIB!r})")
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    psf = torch.from_numpy(orc.synthetic_psf(1, 40, 56, 3, seed=1)).cuda()
    y = torch.rand((1, 40, 56, 3), device="cuda")
    native = ns["NativeADMM"](psf, 1e-6, 1e-5, 4e-5, 1e-4)
    got = native.run(y, 7)
    torch.cuda.synchronize()
    rec = lpa.ADMM(psf)
    rec.set_data(y[0])
    assert torch.equal(got, rec.apply(n_iter=7, disp_iter=None))
    del native


def test_library_reads_the_environment_in_three_places_only():
    """Launch-plan choices travel in lpc_config.options (include/lpc.h); the library itself may look at LPC_OPTIONS
    (process-wide defaults, same syntax), ROCM_PATH (where hipcc lives) and HOME (per-user module cache) -- nothing else,
    outside the LPC_DEBUG_KNOBS build."""
    import re

    csrc = os.path.join(ROOT, "lenslesspicam_amd", "csrc")
    calls = []
    for f in sorted(os.listdir(csrc)):
        debug = False
        for ln in open(os.path.join(csrc, f)):
            if ln.startswith("#ifdef LPC_DEBUG_KNOBS"):
                debug = True
            elif ln.startswith("#endif"):
                debug = False
            elif not debug:
                calls += re.findall(r'getenv\("([A-Z_]+)"\)', ln)
    assert sorted(calls) == ["HOME", "LPC_OPTIONS", "ROCM_PATH"], calls
    for f in sorted(os.listdir(os.path.join(ROOT, "tests"))):
        if f.endswith(".py") and f != "test_abi_and_layout.py":
            src = open(os.path.join(ROOT, "tests", f)).read()
            # (LPC_EMU_* / LPC_SAN_*: which build of the test-only emulator the harness loads -- conftest.py, test_sanitizer.py)
            assert not re.search(r'(environ|setenv)[^\n]*"LPC_(?!EMU_|SAN_)', src), f
