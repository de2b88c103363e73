"""C1 under `rocprofv3 --kernel-trace`: 30 applies of 5 iterations (reset + 5 it + read-out); tools/probe/trace_gaps.py
turns the trace into the per-apply kernel sequence with durations and the idle gaps between dependent kernels."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, lenslesspicam_amd as lpa
H, W, C = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (270, 480, 3)))
n_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 5
opts = sys.argv[5] if len(sys.argv) > 5 else ""
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
psf = torch.rand((1, H, W, C), device=dev, generator=g) ** 12
psf /= psf.norm()
y = torch.rand((H, W, C), device=dev, generator=g)
r = lpa.ADMM(psf, engine_options=opts)
r.set_data(y)
for _ in range(3):
    r.apply(n_iter=n_iter, disp_iter=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    r.apply(n_iter=n_iter, disp_iter=None)
torch.cuda.synchronize()
print(f"{H}x{W}x{C} [{opts}]: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per apply of {n_iter} iterations (wall)")
