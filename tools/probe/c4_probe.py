"""C4 (64 x 270x480x3, ADMM 20 it) on one GPU with the launch-plan options given as argv[1] ("k=v,k=v"): frame-it/s and
the per-kernel HIP-event times -- for same-box A/B runs (tools/c4_ab.sh)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, lenslesspicam_amd as lpa
opts = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
B = 64
psf = torch.rand((1, 270, 480, 3), device=dev, generator=g) ** 12; psf /= psf.norm()
y = torch.rand((B, 270, 480, 3), device=dev, generator=g)
r = lpa.ADMM(psf, engine_options=opts); r.set_data(y[:, None]); r.apply_batch(n_iter=20); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): r.apply_batch(n_iter=20)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
r._handle.profile_enable(True); r.apply_batch(n_iter=20); prof = r._handle.profile_read()
print(f"[{opts}]", round(B * 20 / t), {k: round(v[0], 3) for k, v in prof.items() if v[1]})
