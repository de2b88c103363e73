"""Builds the product library in-tree: hipcc, gfx950 only.  (No JIT cache: the .so must travel
with the source snapshot to the GPU box.)"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "lpc_engine.cpp")
OUT = os.path.join(HERE, "_lib", "liblpc.so")            # float32
OUT_F64 = os.path.join(HERE, "_lib", "liblpc_f64.so")    # same translation unit with -DLPC_DOUBLE


def sources():
    csrc = os.path.join(HERE, "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + [os.path.join(ROOT, "include", "lpc.h")]


def is_stale():
    return any(not os.path.exists(o) or any(os.path.getmtime(f) > os.path.getmtime(o) for f in sources())
               for o in (OUT, OUT_F64))


def build_hip(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    base = [hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-x", "hip", SRC,
            "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc")]
    procs = []
    for out, extra in ((OUT, []), (OUT_F64, ["-DLPC_DOUBLE"])):
        cmd = base + extra + ["-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))     # the two builds run side by side
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return OUT


if __name__ == "__main__":
    build_hip(force=True)
