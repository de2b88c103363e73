"""Builds the product library in-tree: hipcc, gfx950 only.  (No JIT cache: the .so must travel
with the source snapshot to the GPU box.)"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "lpc_engine.cpp")
OUT = os.path.join(HERE, "_lib", "liblpc.so")


def sources():
    csrc = os.path.join(HERE, "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + [os.path.join(ROOT, "include", "lpc.h")]


def is_stale():
    return not os.path.exists(OUT) or any(os.path.getmtime(f) > os.path.getmtime(OUT) for f in sources())


def build_hip(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-x", "hip", SRC,
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build_hip(force=True)
