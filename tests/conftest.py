import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_DIR = os.path.join(ROOT, "tests", "simt_emu")
# LPC_EMU_FLAVOUR=san: the AddressSanitizer + UndefinedBehaviorSanitizer build of the emulator (build_emu.sh --san,
# tests/test_sanitizer.py runs a part of the suite on it in a child process with the ASan runtime preloaded)
EMU_SAN = os.environ.get("LPC_EMU_FLAVOUR", "") == "san"
EMU_BUILD = os.path.join(EMU_DIR, "_build_san" if EMU_SAN else "_build")
EMU_LIB = os.path.join(EMU_BUILD, "liblpc_emu.so")
EMU_LIB_F64 = os.path.join(EMU_BUILD, "liblpc_emu_f64.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")



def _emu_sources():
    csrc = os.path.join(ROOT, "lenslesspicam_amd", "csrc")
    files = [os.path.join(csrc, f) for f in os.listdir(csrc)]
    files += [os.path.join(EMU_DIR, "emu.cpp"), os.path.join(ROOT, "include", "lpc.h")]
    return files


@pytest.fixture(scope="session")
def emu_lib():
    """SIMT-emulator build of the engine's real kernel sources (tests only, see lpc_rt.h)."""
    from lenslesspicam_amd import _native

    libs = (EMU_LIB,) if EMU_SAN else (EMU_LIB, EMU_LIB_F64)       # (the sanitizer flavour is float32 only)
    stale = any(not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in _emu_sources())
                for lib in libs)
    if stale:
        subprocess.check_call(["sh", os.path.join(EMU_DIR, "build_emu.sh")] + (["--san"] if EMU_SAN else []))
    lib = _native.Lib(EMU_LIB)
    lib.f64 = None if EMU_SAN else _native.Lib(EMU_LIB_F64)     # the float64 flavour rides along
    return lib


class Backend:
    def __init__(self, kind, lib, device):
        self.kind, self.lib, self.device = kind, lib, device


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """'emu': the kernel sources on the CPU SIMT emulator (default suite);
    'hip': the product library on cuda:0 (-m gpu).  Both go through the same C ABI and the
    same Python boundary classes."""
    import torch

    from lenslesspicam_amd import recon

    if request.param == "emu":
        lib = request.getfixturevalue("emu_lib")
        dev = torch.device("cpu")
        monkeypatch.setattr(recon, "runtime", lambda dtype="float32": (lib.f64 if dtype == "float64" else lib, dev))
        return Backend("emu", lib, dev)
    lib, dev = recon.runtime()  # raises without a GPU / without the HIP library
    assert lib.backend().startswith("hip")
    return Backend("hip", lib, dev)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
