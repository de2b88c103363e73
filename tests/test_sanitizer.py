"""
SURVEY.md section 5, sanitizer row: the engine's REAL kernel sources under AddressSanitizer + UndefinedBehaviorSanitizer.

`tests/simt_emu/build_emu.sh --san` builds the SIMT-emulator flavour of the library (float32) with
`-fsanitize=address,undefined -fno-sanitize-recover=all`; the library hands the same flags to its own JIT
(LPC_MODULE_EXTRA_DEFS), so the PLAN MODULES -- the compile-time-plan kernels that are the production path -- are
instrumented as well.  The emulator tells ASan about every fibre switch, gives each workgroup an exact-size LDS block and
every device allocation is a plain `malloc` of the exact size: an LDS or global index one element out of range is a report,
and a report is fatal.  The reference's counterpart is the anomaly check its own tests switch on
(`/root/reference/test/test_algos.py:22`, `lensless/recon/utils.py:823`).

The default CPU suite runs a part of the parity suite on that flavour in a child process (the ASan runtime must be the
first library of the process: LD_PRELOAD); `LPC_SAN_FULL=1` runs all of test_parity_small.py, test_norm_scale.py, the
random-shape tests of test_parity_large.py and test_dist.py (about 50 minutes on 16 cores; last full run recorded in
profiles/r05_sanitizer_full.log).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "simt_emu")
SAN_LIB = os.path.join(EMU_DIR, "_build_san", "liblpc_emu.so")

# one case of every kernel family, run-time plans and plan modules: paired and half-length ADMM rows with the X half and
# the sensor-window structure, the tiled image-domain kernel and its form inside the forward rows, the LDS / sequential
# middles (plain and pair-line spectra, odd window offsets, an odd padded height), the gradient-descent family's fused rows (both forms), a random odd shape (LPC_SAN_FULL=1 adds everything
# else: pass A, the register middles, the operator, tiny frames, the world-size-2 gloo tests)
SUBSET = " or ".join([
    "test_admm_matches_reference_golden and admm_24x32x3_tv",
    "test_gd_family_matches_reference_golden and fista_24x32x3",
    "test_admm_half_length_row_kernels and admm_24x32x3_tv and static_plan",
    "test_c4_sequential_middle_on_one_frame",
    "test_pair_line_spectra_odd_window_and_odd_height",
    "test_gd_fused_rows_second_form and shape0",
    "test_tv_half_inside_forward_rows and shape0",
    "test_random_small_shapes_through_plan_modules and emu-0",
])
FILES = ["tests/test_parity_small.py"]
FULL_FILES = ["tests/test_parity_small.py", "tests/test_norm_scale.py", "tests/test_parity_large.py", "tests/test_dist.py"]


def _san_env():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no AddressSanitizer runtime (gcc -print-file-name=libasan.so)")
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": asan,
        # leaks are the interpreter's (and torch's) business; everything else is fatal
        "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=99",
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=98",
        "LPC_EMU_FLAVOUR": "san",
        "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", ""),
    })
    return env


def _build_if_stale():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest

    if not os.path.exists(SAN_LIB) or any(os.path.getmtime(f) > os.path.getmtime(SAN_LIB) for f in conftest._emu_sources()):
        subprocess.check_call(["sh", os.path.join(EMU_DIR, "build_emu.sh"), "--san"])


def test_sanitizer_flavour_reports_an_lds_overrun():
    """The instrumentation is live: a workgroup that reads one byte past its LDS dies with an AddressSanitizer report in
    the --san build (and only there: the plain build reads the slack behind the tile and returns)."""
    env = _san_env()
    _build_if_stale()
    code = "import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); lib.lpc_emu_selftest_lds_overrun(); print('survived')"
    r = subprocess.run([sys.executable, "-c", code, SAN_LIB], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr and "survived" not in r.stdout, (r.returncode, r.stderr[-2000:])
    plain = os.path.join(EMU_DIR, "_build", "liblpc_emu.so")
    if os.path.exists(plain):
        r = subprocess.run([sys.executable, "-c", code, plain], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "survived" in r.stdout


def test_parity_suite_under_address_and_ub_sanitizer():
    """A part of the parity suite (all of it with LPC_SAN_FULL=1) on the sanitizer flavour: green, and no report."""
    env = _san_env()
    _build_if_stale()
    full = os.environ.get("LPC_SAN_FULL", "") not in ("", "0")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"]
    if full:
        cmd += FULL_FILES + ["-k", "not float64 and not f64 and not test_c1_ and not test_c2_ and not test_c5_ and not 760x1014 and not on_the_gpu"]
    else:
        cmd += FILES + ["-k", SUBSET]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=7200 if full else 1500)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert " passed" in r.stdout, tail
