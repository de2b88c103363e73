"""How much of a small-frame ADMM iteration is idle time between dependent kernels?  Run under
`rocprofv3 --kernel-trace`; prints wall time per iteration; the trace gives the busy time."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, lenslesspicam_amd as lpa
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
for (H, W, C) in ((270, 480, 3), (760, 1014, 1)):
    psf = torch.rand((1, H, W, C), device=dev, generator=g) ** 12
    psf /= psf.norm()
    y = torch.rand((H, W, C), device=dev, generator=g)
    r = lpa.ADMM(psf)
    r.set_data(y)
    r.apply(n_iter=100, disp_iter=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.apply(n_iter=100, disp_iter=None)
    torch.cuda.synchronize()
    print(f"{H}x{W}x{C}: {(time.perf_counter() - t0) / 5 / 100 * 1e6:.1f} us per iteration (wall)")
