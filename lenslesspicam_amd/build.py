"""Builds the product library in-tree: hipcc, gfx950 only.  (No JIT cache outside the tree: the .so files must travel
with the source snapshot to the GPU box.)

The engine is split into translation units (csrc/lpc_engine.h) so that the device compiler works on them in
parallel: every ``csrc/*.cpp`` except ``lpc_module.cpp`` is compiled to an object file per flavour (float32, and
float64 with ``-DLPC_DOUBLE``), all jobs side by side, then linked into ``_lib/liblpc.so`` / ``_lib/liblpc_f64.so``.
No relocatable device code is needed: a kernel is always launched from the unit that instantiates it.

``lpc_module.cpp`` is the source of the PLAN MODULES (csrc/lpc_plan.h): the compile-time-plan kernels of one frame
shape, one small shared object per shape under ``_lib/modules/``.  The library compiles a missing module itself on
first use (csrc/lpc_jit.cpp); ``build_modules()`` asks it to do that now -- no GPU needed -- for the shapes of
BASELINE.json's configurations, so that a fresh GPU box starts with them on disk."""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib", "liblpc.so")            # float32
OUT_F64 = os.path.join(HERE, "_lib", "liblpc_f64.so")    # same translation units with -DLPC_DOUBLE
OBJ = os.path.join(HERE, "_lib", "obj")
MODULES = os.path.join(HERE, "_lib", "modules")


def units():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".cpp") and f != "lpc_module.cpp"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "lpc.h")]


def fingerprint():
    """names the sources a library was built from: a plan module is only ever loaded by the library it was built for"""
    h = hashlib.sha1()
    for f in sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(os.environ.get("LPC_EXTRA_DEFS", "").encode())
    return h.hexdigest()[:12]


def sources_crc():
    """CRC-32 the library carries of its own sources (csrc/lpc_jit.cpp: sources_crc recomputes it before it compiles a
    plan module, so that a module can never come from sources edited after the library was built)"""
    import zlib

    crc = 0
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".cpp", ".inc")))
    for path in [os.path.join(CSRC, f) for f in names] + [os.path.join(ROOT, "include", "lpc.h")]:
        crc = zlib.crc32(os.path.basename(path).encode(), crc)
        crc = zlib.crc32(open(path, "rb").read(), crc)
    return crc & 0xFFFFFFFF


FP_FILE = os.path.join(HERE, "_lib", "BUILD_FP")


def is_stale():
    """the libraries are missing, or were built from other sources than the ones in the tree (by content, not by
    modification time: a snapshot copied to another machine must not look stale, an edited header must)"""
    if not (os.path.exists(OUT) and os.path.exists(OUT_F64) and os.path.exists(FP_FILE)):
        return True
    return open(FP_FILE).read().strip() != fingerprint()


def build_hip(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    fp = fingerprint()
    base = [hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-x", "hip",
            "-I", os.path.join(ROOT, "include"), "-I", CSRC, f'-DLPC_SRC_FP="{fp}"', f"-DLPC_SRC_CRC=0x{sources_crc():08x}u"]
    extra_defs = os.environ.get("LPC_EXTRA_DEFS", "").split()    # e.g. -DLPC_DEBUG_KNOBS for timing experiments
    if extra_defs:                                               # ... which the library hands on to its plan modules
        base.append('-DLPC_MODULE_EXTRA_DEFS="' + " ".join(extra_defs) + '"')
    flavours = (("f32", OUT, extra_defs), ("f64", OUT_F64, ["-DLPC_DOUBLE"] + extra_defs))
    for stale in os.listdir(OBJ):                                 # objects of units that no longer exist
        if stale.rsplit(".", 2)[0] + ".cpp" not in {os.path.basename(u) for u in units()}:
            os.remove(os.path.join(OBJ, stale))
    jobs = []
    for tag, _, extra in flavours:
        for src in units():
            obj = os.path.join(OBJ, f"{os.path.basename(src)[:-4]}.{tag}.o")
            cmd = base + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))        # every unit of both flavours side by side
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    for tag, out, _ in flavours:
        objs = [os.path.join(OBJ, f"{os.path.basename(src)[:-4]}.{tag}.o") for src in units()]
        # the soname lets a plan module's DT_NEEDED entry resolve to the copy of the library already in the process
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,-soname,{os.path.basename(out)}"] + objs + \
              ["-o", out, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    open(FP_FILE, "w").write(fp + "\n")
    if os.path.isdir(MODULES):                               # modules of other sources are dead weight in the snapshot
        for f in os.listdir(MODULES):
            if f.startswith("lpcmod_") and f"_{fp}_" not in f:
                os.remove(os.path.join(MODULES, f))
    return OUT


# frame shapes whose plan modules are pre-built: BASELINE.json's configurations C1-C5, the
# profile/*.py frame, and the reference's standard frames beyond them -- RPi-HQ (3040 x 4056, lensless/hardware/sensor.py:76)
# at `downsample` 2 / 4 / 8 (configs/recon/defaults.yaml:20 ships 4) and DiffuserCam-Mirflickr frames (270 x 480) gray and
# RGB at batch 1 / 8 / 16 / 64 -- so that a deployment without a compiler runs them on compile-time plans too.
# (algo: 1 ADMM, 4 FISTA -- the gradient-descent family and the bare operator share one module)
PREBUILT = [
    dict(algo=1, height=3040, width=4056, channels=3), dict(algo=4, height=3040, width=4056, channels=3),   # C2, C3
    dict(algo=1, height=270, width=480, channels=3), dict(algo=4, height=270, width=480, channels=3),       # C1
    dict(algo=1, height=270, width=480, channels=3, batch=64),                                               # C4
    dict(algo=1, height=270, width=480, channels=3, batch=8),                                                # C4 / 8 GPUs
    dict(algo=1, height=1080, width=1920, channels=3, depth=16), dict(algo=4, height=1080, width=1920, channels=3),  # C5
    dict(algo=1, height=760, width=1014, channels=1), dict(algo=4, height=760, width=1014, channels=1),     # profile/*.py
    dict(algo=1, height=1080, width=1920, channels=3),                                                       # one plane of C5
    dict(algo=1, height=1520, width=2028, channels=3), dict(algo=4, height=1520, width=2028, channels=3),   # RPi-HQ / 2
    dict(algo=1, height=760, width=1014, channels=3), dict(algo=4, height=760, width=1014, channels=3),     # RPi-HQ / 4
    dict(algo=1, height=380, width=507, channels=3), dict(algo=4, height=380, width=507, channels=3),       # RPi-HQ / 8
    dict(algo=1, height=270, width=480, channels=3, batch=16),                                               # DiffuserCam RGB
    dict(algo=1, height=270, width=480, channels=1), dict(algo=4, height=270, width=480, channels=1),       # ... gray
    dict(algo=1, height=270, width=480, channels=1, batch=8), dict(algo=1, height=270, width=480, channels=1, batch=16),
    dict(algo=1, height=270, width=480, channels=1, batch=64),
]
PREBUILT_F64 = [dict(algo=1, height=3040, width=4056, channels=3), dict(algo=4, height=3040, width=4056, channels=3)]


def build_modules(verbose=True):
    """Compiles the plan modules of PREBUILT that are not on disk yet (in parallel; ~3 s of hipcc each)."""
    from . import _native

    os.makedirs(MODULES, exist_ok=True)
    jobs = [(_native.Lib(OUT), kw) for kw in PREBUILT] + [(_native.Lib(OUT_F64), kw) for kw in PREBUILT_F64]
    keys = [lib.plan_module(build=False, **kw) for lib, kw in jobs]
    todo = {}                        # several configurations may share one module (C4's batch sizes, C5 and its planes)
    for (lib, kw), key in zip(jobs, keys):
        if key:
            todo.setdefault((lib.path, key), (lib, kw))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:      # ctypes releases the GIL
        list(pool.map(lambda j: j[0].plan_module(build=True, **j[1]), todo.values()))   # (no-op for modules on disk)
    if verbose:
        for (lib, kw), key in zip(jobs, keys):
            print(f"plan module {lib.real} {kw}: {key or '(run-time plans)'}", flush=True)
    return keys


if __name__ == "__main__":
    build_hip(force=True)
    build_modules()
