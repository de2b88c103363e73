"""C5 (16 planes x 1080x1920x3, ADMM) kernel breakdown; tuning knobs via the environment."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, lenslesspicam_amd as lpa
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
psf = torch.rand((16, 1080, 1920, 3), device=dev, generator=g) ** 12
psf /= psf.norm()
y = torch.rand((1080, 1920, 3), device=dev, generator=g)
r = lpa.ADMM(psf)
r.set_data(y)
r.apply(n_iter=10, disp_iter=None)
torch.cuda.synchronize()
t0 = time.perf_counter()
r.apply(n_iter=30, disp_iter=None)
torch.cuda.synchronize()
t = time.perf_counter() - t0
r._handle.profile_enable(True)
r.apply(n_iter=10, disp_iter=None)
prof = r._handle.profile_read()
print({k: os.environ.get(k) for k in ("LPC_SPLIT_N2", "LPC_ROWS_PAIRED", "LPC_ROWS_HALF", "LPC_MID_LDS")},
      round(30 / t, 2), "it/s", {k: round(v[0], 3) for k, v in prof.items() if v[1]})
