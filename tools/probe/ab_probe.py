"""Same-box A/B of launch-plan options (include/lpc.h) on one workload.
usage: ab_probe.py ALGO D H W C B N_ITER REPS "opts_a" "opts_b" ...   (ALGO: admm | fista; opts: "k=v,k=v" or "")
Every option string is run ROUNDS times, interleaved; prints ms per call (best and median) and the kernel table."""
import sys, os, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, lenslesspicam_amd as lpa
algo, D, H, W, C, B, n_iter, reps = sys.argv[1], *(int(v) for v in sys.argv[2:9])
variants = sys.argv[9:] or [""]
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
psf = torch.rand((D, H, W, C), device=dev, generator=g) ** 12; psf /= psf.norm()
y = torch.rand((B, H, W, C), device=dev, generator=g)
cls = lpa.ADMM if algo == "admm" else lpa.FISTA
recs = []
for o in variants:
    r = cls(psf, engine_options=o)
    r.set_data(y[:, None] if B > 1 else y[0])
    recs.append(r)
call = (lambda r: r.apply_batch(n_iter=n_iter)) if B > 1 else (lambda r: r.apply(n_iter=n_iter, disp_iter=None))
times = [[] for _ in variants]
for rnd in range(5):
    for i, r in enumerate(recs):
        call(r); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): call(r)
        torch.cuda.synchronize()
        times[i].append((time.perf_counter() - t0) / reps * 1e3)
for i, r in enumerate(recs):
    r._handle.profile_enable(True); call(r); prof = r._handle.profile_read(); r._handle.profile_enable(False)
    print(f"[{variants[i]}] best {min(times[i]):.3f} ms  median {statistics.median(times[i]):.3f} ms  "
          f"{ {k: round(v[0], 4) for k, v in prof.items() if v[1]} }\n    {r._handle.plan_info()}")
