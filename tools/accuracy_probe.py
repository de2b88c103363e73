"""GPU box: engine (f32) vs oracle f32 vs oracle f64 on a mid-size frame -- whose error is it?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc
torch.set_num_threads(64)
H, W, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = [int(v) for v in sys.argv[4].split(",")]
psf = orc.synthetic_psf(1, H, W, C, seed=0)
scene = orc.synthetic_scene(H, W, C, seed=1)
y = orc.synthetic_measurement(psf, scene)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
o32 = orc.ADMMOracle(psf); o32.set_data(y)
o64 = orc.ADMMOracle(psf, dtype=torch.float64); o64.set_data(y)
rec = lpa.ADMM(torch.from_numpy(psf).cuda()); rec.set_data(torch.from_numpy(y).cuda())
done = 0
rec.reset()
for n in iters:
    t = time.time()
    for _ in range(n - done): o32.step(); o64.step()
    rec.apply(n_iter=n - done, disp_iter=None, reset=False)
    done = n
    Vg = rec._image_est.cpu()
    print(f"it{n}: max|V|={float(o64.V.abs().max()):.3e}  gpu-vs-f64={rel(Vg, o64.V):.2e}  cpu32-vs-f64={rel(o32.V, o64.V):.2e}  gpu-vs-cpu32={rel(Vg, o32.V):.2e}  ({time.time()-t:.0f}s)", flush=True)
