"""Builds the product library in-tree: hipcc, gfx950 only.  (No JIT cache: the .so must travel
with the source snapshot to the GPU box.)

The engine is split into translation units (csrc/lpc_engine.h) so that the device compiler works on them in
parallel: every ``csrc/*.cpp`` is compiled to an object file per flavour (float32, and float64 with
``-DLPC_DOUBLE``), all jobs side by side, then linked into ``_lib/liblpc.so`` / ``_lib/liblpc_f64.so``.
No relocatable device code is needed: a kernel is always launched from the unit that instantiates it."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_lib", "liblpc.so")            # float32
OUT_F64 = os.path.join(HERE, "_lib", "liblpc_f64.so")    # same translation units with -DLPC_DOUBLE
OBJ = os.path.join(HERE, "_lib", "obj")


def units():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".cpp")]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "lpc.h")]


def is_stale():
    return any(not os.path.exists(o) or any(os.path.getmtime(f) > os.path.getmtime(o) for f in sources())
               for o in (OUT, OUT_F64))


def build_hip(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ, exist_ok=True)
    base = [hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-x", "hip",
            "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    extra_defs = os.environ.get("LPC_EXTRA_DEFS", "").split()    # e.g. -DLPC_DEBUG_KNOBS for timing experiments
    flavours = (("f32", OUT, extra_defs), ("f64", OUT_F64, ["-DLPC_DOUBLE"] + extra_defs))
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
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build_hip(force=True)
