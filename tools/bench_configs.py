#!/usr/bin/env python3
"""Times the BASELINE.json configs other than the headline one on ONE GPU (bench.py owns C2).
Prints one JSON object per config: engine it/s (HIP events around apply), and for the small configs the
CPU-oracle it/s on 16 host threads for orientation."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lenslesspicam_amd as lpa
from oracle import lensless_oracle as orc

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def rand_inputs(D, H, W, C, B, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    psf = torch.rand((D, H, W, C), device=dev, generator=g) ** 12
    psf /= psf.norm()
    y = torch.rand((B, H, W, C), device=dev, generator=g)
    return psf, y

out = []
# C1: single DiffuserCam-sized frame, ADMM 5 iterations (profile/admm.py plumbing)
psf, y = rand_inputs(1, 270, 480, 3, 1)
rec = lpa.ADMM(psf); rec.set_data(y[0])
t = timed(lambda: rec.apply(n_iter=5, disp_iter=None), reps=20)
torch.set_num_threads(16)
o = orc.ADMMOracle(psf.cpu().numpy()); o.set_data(y[0].cpu().numpy()); o.apply(5)
t0 = time.perf_counter(); o.apply(5); tc = time.perf_counter() - t0
out.append({"config": "C1 270x480x3 ADMM 5 it (apply incl. reset+form_image)", "ms_per_apply": t * 1e3, "it_per_s": 5 / t,
            "cpu_oracle_it_per_s_16thr": 5 / tc})
t = timed(lambda: rec.apply(n_iter=100, disp_iter=None), reps=5)
out.append({"config": "C1 270x480x3 ADMM 100 it", "ms_per_apply": t * 1e3, "it_per_s": 100 / t})
# C3: FISTA 300 iterations at 12 MP
psf, y = rand_inputs(1, 3040, 4056, 3, 1)
f = lpa.FISTA(psf); f.set_data(y[0])
t = timed(lambda: f.apply(n_iter=300, disp_iter=None), reps=1)
out.append({"config": "C3 3040x4056x3 FISTA 300 it", "s_per_apply": t, "it_per_s": 300 / t})
del f
# C4: 64 DiffuserCam frames, ADMM 20 iterations, all on one GPU (8-GPU sharding = 8 frames/GPU)
for B in (64, 8):
    psf, y = rand_inputs(1, 270, 480, 3, B)
    r = lpa.ADMM(psf); r.set_data(y[:, None])
    t = timed(lambda: r.apply_batch(n_iter=20), reps=3)
    r._handle.profile_enable(True)
    r.apply_batch(n_iter=20)
    prof = r._handle.profile_read()
    r._handle.profile_enable(False)
    kern = {k: {"ms": round(v[0], 4), "GBps": round(r._handle.kernel_bytes(i) / (v[0] * 1e-3) / 1e9, 0)}
            for i, (k, v) in enumerate(prof.items()) if v[1]}
    out.append({"config": f"C4 batch {B} x 270x480x3 ADMM 20 it on 1 GPU", "ms_per_batch": t * 1e3,
                "frame_it_per_s": B * 20 / t, "kernels": kern})
    del r
# C5: 16 depth planes 1080x1920x3, ADMM 50 iterations
psf, y = rand_inputs(16, 1080, 1920, 3, 1)
r = lpa.ADMM(psf); r.set_data(y[0])
t = timed(lambda: r.apply(n_iter=50, disp_iter=None), reps=1)
out.append({"config": "C5 16 planes x 1080x1920x3 ADMM 50 it", "s_per_apply": t, "it_per_s": 50 / t,
            "hbm_GB": r._handle.workspace_bytes() / 1e9})
for o_ in out:
    print(json.dumps(o_))
