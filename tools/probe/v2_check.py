"""GPU check of the second form of the gradient-descent fused rows (option gd_v2) against the first form and the float64
oracle: python tools/probe/v2_check.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


dev = torch.device("cuda", 0)
for (D, H, W, C), kind, cls, n in [((1, 5, 512, 1), "fista", lpa.FISTA, 7), ((1, 5, 512, 1), "fista", lpa.FISTA, 1), ((2, 4, 512, 3), "fista", lpa.FISTA, 7),
                                   ((1, 5, 512, 3), "nesterov", lpa.NesterovGradientDescent, 7), ((1, 3, 512, 1), "vanilla", lpa.GradientDescent, 7),
                                   ((1, 3, 4092, 1), "fista", lpa.FISTA, 7), ((1, 64, 4056, 3), "fista", lpa.FISTA, 7)]:
    rng = np.random.default_rng(W + C)
    psf = orc.synthetic_psf(D, H, W, C, seed=5)
    y = rng.random((H, W, C), dtype=np.float32)
    outs = []
    for v2 in (0, 1):
        rec = cls(torch.from_numpy(psf).to(dev), engine_options={"gd_v2": v2, "jit_min_points": 0})
        rec.set_data(torch.from_numpy(y).to(dev))
        outs.append(rec.apply(n_iter=n, disp_iter=None).detach().cpu().numpy().copy())
    o = orc.GDOracle(psf, kind=kind, dtype=torch.float64); o.set_data(y); ref = o.apply(n)
    print((D, H, W, C), kind, n, "v2-v1 %.3g  v1-orc %.3g  v2-orc %.3g" % (rel(outs[1], outs[0]), rel(outs[0], ref), rel(outs[1], ref)), flush=True)
