"""
Plan-module housekeeping (csrc/lpc_jit.cpp), on the emulator build of the same sources (g++ instead of hipcc):
threads of one process building one module at once, "only if not on disk", the bounded module directory, unloading of
modules nobody uses, paths with separators in the option string, and the fallback to the run-time plans when there is
no compiler (identical results, one warning).  The GPU leg of the last one is in test_parity_large.py.
"""
import os
import time
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import lenslesspicam_amd as lpa
from lenslesspicam_amd import _native

from test_parity_small import engine_opts


def mods(d):
    return sorted(f for f in os.listdir(d) if f.endswith(".so")) if os.path.isdir(d) else []


def test_threads_of_one_process_build_one_module_at_once(emu_lib, tmp_path):
    """build.py compiles PREBUILT from a thread pool, and two configurations may map to ONE module key (C4's batch
    sizes, C5 and one of its planes): every call gets a private temporary (pid + thread + serial)."""
    d = str(tmp_path / "my modules, v1")                     # separators of the option syntax inside the path
    cfgs = [dict(algo=1, height=20, width=44, channels=1, batch=b, options={"jit_min_points": 0, "module_dir": d})
            for b in (1, 2, 3, 4)]
    keys = {emu_lib.plan_module(build=False, **c) for c in cfgs}
    assert len(keys) == 1 and "" not in keys
    with ThreadPoolExecutor(max_workers=4) as pool:
        got = list(pool.map(lambda c: emu_lib.plan_module(build=True, **c), cfgs))
    assert set(got) == keys
    assert len(mods(d)) == 1 and not [f for f in os.listdir(d) if ".tmp" in f], os.listdir(d)
    # "compile it now if it is not on disk" (include/lpc.h): a second call leaves the file alone
    path = os.path.join(d, mods(d)[0])
    before = os.stat(path).st_mtime_ns
    time.sleep(0.05)
    emu_lib.plan_module(build=True, **cfgs[0])
    assert os.stat(path).st_mtime_ns == before


def test_module_directory_is_bounded(emu_lib, tmp_path):
    """option module_max: the directory a library writes its modules to keeps the most recently used ones"""
    d = str(tmp_path / "m")
    shapes = [(20, 44), (20, 52), (24, 44), (24, 60)]
    for i, (h, w) in enumerate(shapes):
        emu_lib.plan_module(build=True, algo=1, height=h, width=w, channels=1,
                            options={"jit_min_points": 0, "module_dir": d, "module_max": 2})
        assert len(mods(d)) == min(i + 1, 2), mods(d)
        time.sleep(0.02)
    newest = emu_lib.plan_module(build=False, algo=1, height=24, width=60, channels=1,
                                 options={"jit_min_points": 0, "module_dir": d})
    assert any(newest in f for f in mods(d))
    open(os.path.join(d, "lpcmod_emu_000000000000_stale.so"), "w").write("x")     # a module of OTHER sources goes first
    emu_lib.plan_module(build=True, algo=1, height=28, width=60, channels=1,
                        options={"jit_min_points": 0, "module_dir": d, "module_max": 3})
    assert len(mods(d)) == 3 and not any("stale" in f for f in mods(d)), mods(d)


def test_unused_modules_are_unloaded_and_come_back(backend, monkeypatch, tmp_path):
    """option module_loaded_max: once no handle uses a module, it is dlclose()d when more than that many are loaded --
    and loads again from disk when the shape returns (same results)."""
    if backend.kind != "emu":
        pytest.skip("emulator leg; the GPU leg is test_parity_large.py::test_modules_unload_on_the_gpu")
    engine_opts(monkeypatch, jit_min_points=0, module_dir=str(tmp_path / "m"), module_loaded_max=1)
    rng = np.random.default_rng(5)

    def run(h, w):
        psf = torch.from_numpy(rng.random((1, h, w, 1), dtype=np.float32) ** 4)
        y = torch.from_numpy(rng.random((h, w, 1), dtype=np.float32))
        rec = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
        assert "plan module" in rec._handle.plan_info()
        rec.set_data(y)
        out = rec.apply(n_iter=3, disp_iter=None)
        rec._handle.close()
        return psf, y, out

    a = run(20, 44)
    run(20, 52)
    run(24, 44)                                  # the first two are unloaded by now
    rec = lpa.ADMM(a[0], tau=2e-6, mu2=1e-4)     # ... and the first one comes back from disk
    assert "plan module" in rec._handle.plan_info()
    rec.set_data(a[1])
    assert torch.equal(rec.apply(n_iter=3, disp_iter=None), a[2])
    assert len(mods(str(tmp_path / "m"))) == 3


def test_no_compiler_falls_back_to_run_time_plans_with_one_warning(backend, monkeypatch, tmp_path):
    """A box without hipcc / g++: the handle runs the run-time plans, says so, warns once -- same results."""
    if backend.kind != "emu":
        pytest.skip("emulator leg; the GPU leg is test_parity_large.py::test_no_compiler_on_the_gpu")
    rng = np.random.default_rng(6)
    psf = torch.from_numpy(rng.random((1, 22, 36, 3), dtype=np.float32) ** 4)
    y = torch.from_numpy(rng.random((22, 36, 3), dtype=np.float32))
    # jit=0 and an empty directory: the emulator's compiler is g++ from PATH, so "no compiler" is modelled by jit=0 here
    # (first: the process keeps a loaded module for every later handle of the shape)
    engine_opts(monkeypatch, jit_min_points=0, module_dir=str(tmp_path / "without"), jit=0)
    monkeypatch.setattr(_native, "_warned", set())
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        rec = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
        rec2 = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
    mine = [w for w in caught if "run-time plans" in str(w.message)]
    assert len(mine) == 1, [str(w.message) for w in caught]
    info = rec._handle.plan_info()
    assert "run-time plans (" in info and "jit=0" in info and "plan module" not in info, info
    assert rec._handle.fallback_reason() and rec2._handle.fallback_reason()
    rec.set_data(y)
    got = rec.apply(n_iter=5, disp_iter=None)
    assert mods(str(tmp_path / "without")) == []
    # a later handle that MAY compile is not held to the earlier failure
    engine_opts(monkeypatch, jit_min_points=0, module_dir=str(tmp_path / "with"), jit=1)
    ref = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
    assert "plan module" in ref._handle.plan_info()
    ref.set_data(y)
    want = ref.apply(n_iter=5, disp_iter=None)
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-6


def test_option_values_with_separators():
    assert _native._option_value("module_dir", "/tmp/a b,c;d%") == "/tmp/a%20b%2Cc%3Bd%25"
    with pytest.raises(ValueError):
        _native._option_value("row_rad", "8,8")
    assert _native._option_value("hv_full", True) == "1"


def test_a_module_file_that_does_not_load_is_replaced(backend, monkeypatch, tmp_path):
    """A file under the module's name that is not a loadable module (truncated by a neighbour that died mid-write, ...):
    found on disk, it fails dlopen -- it is removed and, when the handle may compile, rebuilt once; a jit=0 handle falls
    back to the run-time plans and leaves no wreck behind for the next one."""
    if backend.kind != "emu":
        pytest.skip("emulator leg (the logic is lpc_jit.cpp's, identical in both builds)")
    d = tmp_path / "m"
    d.mkdir()
    cfg = dict(algo=1, height=26, width=44, channels=1)
    key = backend.lib.plan_module(build=False, options={"jit_min_points": 0, "module_dir": str(d)}, **cfg)
    assert key
    # the file name a module of this key carries: build it once elsewhere to learn it
    ref_dir = tmp_path / "ref"
    backend.lib.plan_module(build=True, options={"jit_min_points": 0, "module_dir": str(ref_dir)}, **cfg)
    name = mods(str(ref_dir))[0]
    (d / name).write_bytes(b"\x7fELF not really")
    rng = np.random.default_rng(8)
    psf = torch.from_numpy(rng.random((1, 26, 44, 1), dtype=np.float32) ** 4)
    engine_opts(monkeypatch, jit_min_points=0, module_dir=str(d), jit=0)
    monkeypatch.setattr(_native, "_warned", set())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rec = lpa.ADMM(psf)
    assert "run-time plans (" in rec._handle.plan_info() and "dlopen" in rec._handle.plan_info()
    assert mods(str(d)) == []                                   # the wreck is gone
    engine_opts(monkeypatch, jit_min_points=0, module_dir=str(d), jit=1)
    (d / name).write_bytes(b"\x7fELF not really")               # again, and this handle may compile: replaced in place
    rec = lpa.ADMM(psf)
    assert "plan module " + key in rec._handle.plan_info(), rec._handle.plan_info()
    assert mods(str(d)) == [name] and os.path.getsize(d / name) > 10000
