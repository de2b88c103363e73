"""What the HIP events around every hot-loop kernel cost a timed call: the same solver, 100-iteration calls, events off /
on, interleaved.  usage: event_overhead.py [admm|fista]"""
import sys, os, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, lenslesspicam_amd as lpa
algo = sys.argv[1] if len(sys.argv) > 1 else "admm"
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
H, W, C = 3040, 4056, 3
psf = torch.rand((1, H, W, C), device=dev, generator=g) ** 12; psf /= psf.norm()
y = torch.rand((H, W, C), device=dev, generator=g)
r = (lpa.ADMM if algo == "admm" else lpa.FISTA)(psf)
r.set_data(y)
n_iter = 100
res = {0: [], 1: []}
r.apply(n_iter=n_iter, disp_iter=None); torch.cuda.synchronize()
for rnd in range(4):
    for on in (0, 1):
        r._handle.profile_enable(bool(on))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.apply(n_iter=n_iter, disp_iter=None)
        torch.cuda.synchronize(); res[on].append((time.perf_counter() - t0) * 1e3)
        r._handle.profile_enable(False)
for on in (0, 1):
    print(f"{algo} events {'on ' if on else 'off'}: best {min(res[on]):.2f} ms  median {statistics.median(res[on]):.2f} ms per {n_iter} iterations")
