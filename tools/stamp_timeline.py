"""Where does an iteration of a SMALL frame spend its time?  (profiles/r05_notes.md section 5)

On the GPU box: builds the workload's plan module a second time with -DLPC_STAMP (lpc_rt.h: lane 0 of every workgroup
writes the 100-MHz real-time counter at kernel entry, behind every barrier and -- stores acknowledged -- at exit), runs the
workload through it and prints, for the LAST iteration of the call, the three module kernels on one time axis:
first / last workgroup entry, first / last exit, and the median workgroup's intervals between its stamps.  The image-domain
kernel lives in the core library and is not stamped: it is the gap between the inverse rows of one iteration and the
forward rows of the next (second table: the last TWO iterations need n_iter >= 2 and come from the kernel trace instead).

usage: stamp_timeline.py D H W C B N_ITER [engine options]"""
import ctypes, os, re, statistics, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
KERNELS, WGS, SLOTS = 4, 4096, 32
NAMES = {1: "forward rows (+X half)", 2: "column middle", 3: "inverse rows"}


def make(D, H, W, C, B, opts):
    import torch, lenslesspicam_amd as lpa
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(0)
    psf = torch.rand((D, H, W, C), device=dev, generator=g) ** 12
    psf /= psf.norm()
    y = torch.rand((B, H, W, C), device=dev, generator=g)
    r = lpa.ADMM(psf, engine_options=opts)
    r.set_data(y[:, None] if B > 1 else y[0])
    return r


def timed(r, B, n_iter, reps=50):
    import torch
    call = (lambda: r.apply_batch(n_iter=n_iter)) if B > 1 else (lambda: r.apply(n_iter=n_iter, disp_iter=None))
    for _ in range(5): call()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


def main():
    D, H, W, C, B, n_iter = (int(v) for v in sys.argv[1:7])
    opts = sys.argv[7] if len(sys.argv) > 7 else ""
    if os.environ.get("LPC_STAMP_CHILD") != "1":
        r = make(D, H, W, C, B, opts)
        info = r._handle.plan_info()
        key = re.search(r"plan module (\S+)", info).group(1)
        print("plan:", info)
        print("unstamped: %.4f ms per call of %d iterations" % (timed(r, B, n_iter), n_iter))
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "knock_modules.py"), key, "stamp:-DLPC_STAMP=1"], text=True)
        env = dict(os.environ, LPC_STAMP_CHILD="1", LPC_STAMP_SO=out.strip().splitlines()[-1])
        sys.exit(subprocess.call([sys.executable] + sys.argv, env=env))
    so = os.environ["LPC_STAMP_SO"]
    r = make(D, H, W, C, B, (opts + "," if opts else "") + "module_dir=" + os.path.dirname(so))
    assert os.path.basename(so)[len("lpcmod_hip_"):-3].split("_", 1)[1] in r._handle.plan_info()
    print("stamped:   %.4f ms per call of %d iterations" % (timed(r, B, n_iter), n_iter))
    mod = ctypes.CDLL(so)
    buf = (ctypes.c_ulonglong * (KERNELS * WGS * SLOTS))()
    assert mod.lpc_module_stamps(buf, ctypes.c_size_t(ctypes.sizeof(buf))) == 0
    rec = {}
    for k in NAMES:
        rows = []
        for w in range(WGS):
            o = (k * WGS + w) * SLOTS
            n = int(buf[o])
            if n: rows.append([buf[o + 1 + i] for i in range(min(n, SLOTS - 1))])
        if rows: rec[k] = rows
    t0 = min(min(r_[0] for r_ in rows) for rows in rec.values())
    us = lambda t: (t - t0) / 100.0
    print("\nlast iteration, microseconds after the first workgroup of the forward rows started (100-MHz counter):")
    print("| kernel | workgroups | first in | last in | first out | last out | median workgroup: entry -> [barriers] -> exit |")
    print("|---|---|---|---|---|---|---|")
    for k, rows in sorted(rec.items(), key=lambda kv: min(r_[0] for r_ in kv[1])):
        # a kernel may hold several kinds of workgroup (array 0 / array 1 of the forward rows): group by stamp count
        kinds = {}
        for r_ in rows: kinds.setdefault(len(r_), []).append(r_)
        chains = []
        for n, rs in sorted(kinds.items()):
            med = [statistics.median((r_[i + 1] - r_[i]) / 100.0 for r_ in rs) for i in range(n - 1)]
            chains.append("%d x %d stamps: " % (len(rs), n) + " ".join("%.2f" % m for m in med)
                          + " = %.2f" % statistics.median((r_[-1] - r_[0]) / 100.0 for r_ in rs))
        print("| %s | %d | %.2f | %.2f | %.2f | %.2f | %s |" % (
            NAMES[k], len(rows), us(min(r_[0] for r_ in rows)), us(max(r_[0] for r_ in rows)),
            us(min(r_[-1] for r_ in rows)), us(max(r_[-1] for r_ in rows)), "; ".join(chains)))


main()
