"""Timing-only variants of ONE plan module (lpc_gd_v2_kernels.h: LPC_V2_KNOCK_MASK), compiled here for gfx950 into
_ab_x/knock<mask>/ under the module's regular file name; on the GPU box: option module_dir=_ab_x/knock<mask> (one process
per variant: the process cache is keyed by file name).   python tools/knock_modules.py <plan-module key> 1 2 3 4 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from lenslesspicam_amd import build
import module_asm

key, masks = sys.argv[1], sys.argv[2:]      # a mask may also be a -D list: "tag:-DLPC_V2_TC_RESID=0,-DX=1"
csrc = os.path.join(ROOT, "lenslesspicam_amd", "csrc")
lib = build.OUT_F64 if key.startswith("f64") else build.OUT
fp = open(os.path.join(os.path.dirname(lib), "BUILD_FP")).read().strip().split()[0] if os.path.exists(os.path.join(os.path.dirname(lib), "BUILD_FP")) else build.fingerprint()
for m in masks:
    extra = ["-DLPC_V2_KNOCK_MASK=" + m]
    if ":" in m:
        m, defs = m.split(":", 1)
        extra = defs.split(",")
    d = os.path.join(ROOT, "_ab_x", "knock" + m)
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "lpcmod_hip_%s_%s.so" % (fp, key))
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-x", "hip",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, '-DLPC_SRC_FP="%s"' % fp] + extra + module_asm.defines(key) + [
               os.path.join(csrc, "lpc_module.cpp"), "-x", "none", lib, "-o", out]
    subprocess.check_call(cmd)
    print(out)
