"""gfx950 assembly of one plan module: python tools/module_asm.py <plan-module key> [out.s] [csrc dir]
(the key as lpc_plan_info() / build.py print it, e.g. f32_admm_rp960r8.8.5.3t1w128x8sx_ms540r6.10.9t8w256x17m4);
feed the listing to tools/instmix.py.  LPC_ASM_EXTRA="-DX=1 ...": more compiler arguments."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def defines(key):
    parts = key.split("_")
    d = ["-DLPC_MOD_FAMILY=%d" % {"admm": 1, "gd": 2}[parts[1]]]
    if parts[0] == "f64":
        d.append("-DLPC_DOUBLE")
    row = passa = mid = None
    for p in parts[2:]:
        if p[0] == "r":
            row = p
        elif p[0] == "a":
            passa = p
        elif p[0] == "m":
            mid = p
    fft = r"(\d+)r([\d.]+)t(\d+)w(\d+)x(\d+)"
    if row:
        m = re.fullmatch(r"r([hp])" + fft + r"([szh]?)(x?)", row)
        d += ["-DLPC_MOD_ROW_KIND=%d" % (1 if m[1] == "h" else 2), "-DLPC_MOD_ROW_RAD=" + m[3].replace(".", ","),
              "-DLPC_MOD_ROW_NT=" + m[5], "-DLPC_MOD_ROW_EM=" + m[6], "-DLPC_MOD_ROW_SK=%d" % {"": 0, "s": 1, "z": 2, "h": 3}[m[7]],
              "-DLPC_MOD_ROW_X=%d" % bool(m[8])]
    else:
        d.append("-DLPC_MOD_ROW_KIND=0")
    if passa:
        m = re.fullmatch("a" + fft, passa)
        d += ["-DLPC_MOD_PASSA=1", "-DLPC_MOD_PASSA_RAD=" + m[2].replace(".", ","), "-DLPC_MOD_PASSA_T=" + m[3],
              "-DLPC_MOD_PASSA_NT=" + m[4], "-DLPC_MOD_PASSA_EM=" + m[5]]
    else:
        d.append("-DLPC_MOD_PASSA=0")
    if mid:
        m = re.fullmatch(r"m([ps])" + fft + r"m(\d+)(g?)(p?)(L?)([cr]?)", mid)
        d += ["-DLPC_MOD_MID_KIND=%d" % (1 if m[1] == "p" else 2), "-DLPC_MOD_MID_RAD=" + m[3].replace(".", ","),
              "-DLPC_MOD_MID_T=" + m[4], "-DLPC_MOD_MID_NT=" + m[5], "-DLPC_MOD_MID_EM=" + m[6], "-DLPC_MOD_MID_MINW=" + m[7],
              "-DLPC_MOD_MID_TWG=%d" % bool(m[8]), "-DLPC_MOD_MID_PRE=%d" % bool(m[9]), "-DLPC_MOD_SLAY=%d" % bool(m[10]), "-DLPC_MOD_MID_PC=%d" % {"": 0, "c": 1, "r": 2}[m[11]]]
    else:
        d.append("-DLPC_MOD_MID_KIND=0")
    return d


def main():
    key = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/%s.s" % key
    csrc = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "lenslesspicam_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-x", "hip", "-S", "--cuda-device-only",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, '-DLPC_SRC_FP="asm"'] + defines(key) + [
               os.path.join(csrc, "lpc_module.cpp"), "-o", out] + os.environ.get("LPC_ASM_EXTRA", "").split()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    print(out)


if __name__ == "__main__":
    main()
