"""Does a batch run faster as S independent sub-batches on S HIP streams?  (frames never couple: the latency-bound middle of
one sub-batch could overlap the bandwidth-bound rows of another.)
usage: two_streams.py B N_ITER REPS   -- C4's frame shape; prints ms per call for 1, 2, 4 streams."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import lenslesspicam_amd as lpa

B, n_iter, reps = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
psf = torch.rand((1, 270, 480, 3), device=dev, generator=g) ** 12
psf /= psf.norm()
y = torch.rand((B, 270, 480, 3), device=dev, generator=g)


def build(nstreams, per_iter):
    recs, streams = [], []
    for s in range(nstreams):
        r = lpa.ADMM(psf)
        r.set_data(y[s * B // nstreams:(s + 1) * B // nstreams, None])
        recs.append(r)
        streams.append(torch.cuda.Stream())

    def call():
        torch.cuda.synchronize()
        if per_iter:      # launches interleaved iteration by iteration (one host thread feeds all streams)
            for r, st in zip(recs, streams):
                with torch.cuda.stream(st):
                    r.reset()
            for _ in range(n_iter):
                for r, st in zip(recs, streams):
                    with torch.cuda.stream(st):
                        r._iterate(1)
            outs = []
            for r, st in zip(recs, streams):
                with torch.cuda.stream(st):
                    outs.append(r._form_image())
        else:
            outs = []
            for r, st in zip(recs, streams):
                with torch.cuda.stream(st):
                    outs.append(r.apply_batch(n_iter=n_iter))
        torch.cuda.synchronize()
        return outs
    return call, recs


ref = None
for ns, per_iter in ((1, False), (2, False), (4, False), (8, False), (2, True)):
    call, recs = build(ns, per_iter)
    out = call()
    full = torch.cat([o if o.dim() == 5 else o[None] for o in out], 0)
    if ref is None:
        ref = full
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    same = bool(torch.equal(full.reshape(ref.shape), ref)) if full.numel() == ref.numel() else None
    print(f"{ns} stream(s), {'iteration-interleaved' if per_iter else 'whole calls'}: best {min(ts):.3f} ms  median {sorted(ts)[2]:.3f} ms "
          f"({B * n_iter / min(ts) * 1e3:.0f} frame-it/s); equal to one stream: {same}; {recs[0]._handle.plan_info().split(';')[2]}")
    del recs, call
    torch.cuda.empty_cache()
