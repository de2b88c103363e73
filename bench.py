#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X deconvolution engine.

Metric (BASELINE.json): ADMM iterations/s at 4056x3040x3, 100 iterations (+ PSNR delta vs ref).
Workload C2 (BASELINE.json configs[1]): one 12-MP RPi-HQ frame (3040 x 4056 x 3, padded FFT frame
6144 x 8192) per GPU, ADMM-TV, 100 iterations.  A "step" is one full 100-iteration
reconstruction (`apply(n_iter=100)`) of one frame per GPU with inputs resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: frames are independent, so rank r reconstructs its own frame (weak scaling, no
collective inside the loop); one RCCL all-gather of the final images closes each step, as the
path would do when a batch is sharded over the node.  value = total iterations of all ranks /
max-over-ranks wall time.

Rank 0 prints ONE JSON line.  On top of the driver's contract it carries:
  roofline      ADMM prox / dual-update kernel (the TV / W half of the image-domain work, 8R per launch: reads V,
                eta0, eta1, rho, writes eta0, eta1, rho, r_sp -- the duals travel half-applied between the iterations
                of a call, so V_old is read by the first launch of a call only (9R: 1 launch in 100; the figure is
                priced at 8R) -- DESIGN.md section 4; the X half rides in the
                forward row kernel, listed under `kernels`) / mean launch duration from HIP events recorded inside the
                timed region; `traffic` = HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/traffic.json)
                (only this kernel's launches are bracketed there: events around all of them cost 1.4-3.4 % of the rate)
  kernels       every kernel of the iteration: mean ms, algorithmic GB, GB/s, and the PMC traffic / algorithmic ratio
                (rows other than the roofline kernel's: one extra step after the timed region, all launches bracketed)
  cpu_baseline  the CPU oracle (torch-CPU float32 restatement of the reference, kind "port")
                timed on this host for a bounded sample of the same workload
  parity        the timed call's own output (100 iterations at 12 MP) against the samples the REFERENCE produced on the same
                closed-form inputs (tests/golden/longrun_c2.npz, made by tests/golden/gen_longrun.py from the imported
                reference in float64 and float32): distance to the reference's float64 run, the reference's own float32
                distance to it, PSNR delta vs the scene.  No CPU solver runs for it (tests/test_longrun_pins.py asserts the
                same, and more, in the GPU suite).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak


class DeviceRuntime:
    """Everything bench.py asks of the device besides the engine itself.  Production: HIP (`torch.cuda`, RCCL through the
    "nccl" backend).  LPC_BENCH_BACKEND=emu is TEST INFRASTRUCTURE (tests/test_bench_multirank.py): the same main(), step(),
    wait_gather(), rank_stats() and JSON line on CPU tensors, the SIMT-emulator build of the kernels and a gloo group, so
    that the N > 1 branch -- which no round could run on hardware (one GPU per box) -- is executed before the driver's
    8-GPU node executes it.  It never produces a benchmark number: the emitted line says "backend": "simt-emu"."""

    def __init__(self):
        self.emu = os.environ.get("LPC_BENCH_BACKEND", "") == "emu"

    def device(self, local):
        if self.emu:
            from lenslesspicam_amd import _native, recon

            lib = _native.Lib(os.path.join(ROOT, "tests", "simt_emu", "_build", "liblpc_emu.so"))
            recon.runtime = lambda dtype="float32": (lib, torch.device("cpu"))
            return torch.device("cpu")
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def init_group(self, dist, dev):
        if self.emu:
            dist.init_process_group("gloo")
        else:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (see task environment notes)
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        warm = torch.ones(1, device=dev)
        dist.all_reduce(warm)                            # create the communicator outside any timed region
        self.sync()

    def sync(self):
        if not self.emu:
            torch.cuda.synchronize()

    def empty_cache(self):
        if not self.emu:
            torch.cuda.empty_cache()

    def timed_ms(self, fn, reps):
        """mean milliseconds of `fn` over `reps` calls: device events on the current stream (wall clock on the emulator)"""
        if self.emu:
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t0) * 1e3 / reps
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps

    @property
    def name(self):
        return "simt-emu" if self.emu else "hip-gfx950"


RT = DeviceRuntime()


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-iter", type=int, default=100)
    ap.add_argument("--height", type=int, default=3040)
    ap.add_argument("--width", type=int, default=4056)
    ap.add_argument("--algo", default="admm", choices=["admm", "fista"])
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"],
                    help="float64 runs the second build of the engine (liblpc_f64.so); the headline metric is float32")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5-planes"],
                    help="c2 (default, the headline metric): one 12-MP frame per GPU.  c4: BASELINE config 4, a "
                         "batch of 64 DiffuserCam frames (270x480x3) block-sharded over the ranks, ADMM 20 it, one "
                         "all-gather (strong scaling; reported separately, never as the headline value).  c5-planes: "
                         "BASELINE config 5, ONE frame against a 16-plane depth stack, its 48 (plane, channel) units "
                         "sharded over the ranks (PlaneShardedReconstructor), one all-gather (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3,
                    help="timed CPU-oracle iterations (SURVEY 8d: >= 3); two more are spent choosing the thread count")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short C1 / C3 / C4 / C5 legs reported under 'other_configs'")
    ap.add_argument("--test-shape", type=lambda v: tuple(int(x) for x in v.split(",")), default=None,
                    help="tests only (LPC_BENCH_BACKEND=emu): B,D,H,W,n_iter replacing the sizes of --config c4 / c5-planes")
    return ap.parse_args()


def synth_inputs(H, W, C, seed, device):
    """Closed-form inputs (tests/golden/longrun_inputs.py; exact float32 operations, the same bits on every machine):
    PSF = uniform**12 scaled to unit energy (lensless/utils/io.py:375), measurement = broad bumps + sensor
    noise, clipped and divided by its maximum (io.py:196-197), scene = what PSNR is quoted against.  Rank r (seed r)
    gets its own measurement; seed 0 is the frame the reference's 100-iteration run is on file for."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import longrun_inputs as li

    psf = torch.from_numpy(li.psf12(1, H, W, C, seed=0)).to(device)
    y = torch.from_numpy(li.measurement(H, W, C, seed=seed)).to(device)
    return psf, li.scene(H, W, C), y


def reference_parity(out, scene, H, W, n_iter, algo):
    """the engine's output against the samples of the reference's own run at this size and length, if on file"""
    if (H, W) != (3040, 4056):
        return None
    tag, name = ("c2", "admm") if algo == "admm" else ("c3", "fista")
    path = os.path.join(ROOT, "tests", "golden", f"longrun_{tag}.npz")
    if not os.path.exists(path):
        return None
    import longrun_inputs as li

    fx = np.load(path)
    if n_iter not in [int(v) for v in fx[f"{name}_iters"]]:
        return None
    img = out.detach().cpu().numpy()[0]
    r64c, r64l, r64s = (fx[f"{name}_f64_it{n_iter}_{k}"] for k in ("crops", "lattice", "stats"))
    r32c, r32l = (fx[f"{name}_f32_it{n_iter}_{k}"] for k in ("crops", "lattice"))
    crops, lat = li.samples(img)
    top = float(r64s[2])
    d32 = max(np.abs(crops - r64c).max(), np.abs(lat - r64l).max()) / top
    dref = max(np.abs(r32c - r64c).max(), np.abs(r32l - r64l).max()) / top
    st = li.stats(img, scene)
    return {"against": f"tests/golden/longrun_{tag}.npz: the imported reference, torch-CPU, {n_iter} iterations on these inputs "
                       "(8 crops of 32x32x3 + a stride-61 lattice over the frame)",
            "engine_f32_vs_reference_f64": float(d32), "reference_f32_vs_reference_f64": float(dref),
            "psnr_db": {"engine": float(st[4]), "reference_f64": float(r64s[4]),
                        "reference_f32": float(fx[f"{name}_f32_it{n_iter}_stats"][4])},
            "psnr_delta_db": float(st[4] - r64s[4])}


def rank_stats(dist, dev, elapsed, units_per_rank):
    """max-over-ranks time (the contract's clock) + every rank's own rate: min / max say how even the ranks ran"""
    if not dist:
        return elapsed, {"rccl_world": 1, "per_rank_units_per_s_min": units_per_rank / elapsed,
                         "per_rank_units_per_s_max": units_per_rank / elapsed}
    world = dist.get_world_size()
    mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    every = torch.empty(world, device=dev, dtype=torch.float64)
    dist.all_gather_into_tensor(every, mine)
    every = every.cpu().tolist()
    return max(every), {"rccl_world": world, "per_rank_units_per_s_min": units_per_rank / max(every),
                        "per_rank_units_per_s_max": units_per_rank / min(every), "per_rank_s": [round(v, 4) for v in every]}


def timed_steps(args, dist, step):
    for _ in range(max(args.warmup, 1)):
        step()
    if dist:
        dist.barrier()
    RT.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    RT.sync()
    if dist:
        dist.barrier()
    return time.perf_counter() - t0, out


def run_c4(args, rank, world, dev, dist):
    """BASELINE config 4: 64 frames 270x480x3 sharing one PSF, ADMM 20 iterations, frames block-sharded over
    the ranks (lenslesspicam_amd.dist), ONE all-gather of the results per step.  Strong scaling."""
    import lenslesspicam_amd as lpa
    B, H, W, C, n_iter = 64, 270, 480, 3, 20
    if args.test_shape:      # (tests only: B,D,H,W,n_iter of a frame the emulator finishes in seconds)
        B, _, H, W, n_iter = args.test_shape
    g = torch.Generator(device=dev).manual_seed(0)
    psf = torch.rand((1, H, W, C), device=dev, generator=g) ** 12
    psf /= psf.norm()
    frames = torch.rand((B, H, W, C), device=dev, generator=g)       # same on every rank (same seed)

    from lenslesspicam_amd.dist import ShardedReconstructor, shard_bounds

    # the solver (handle, PSF spectrum, workspace) is built ONCE; reuse_output: the gathered batch is handed out in the
    # receive buffer itself (valid until the call after the next), as a steady-state pipeline would run it
    sharded = ShardedReconstructor(lpa.ADMM, psf, reuse_output=True)
    elapsed, out = timed_steps(args, dist, lambda: sharded(frames, n_iter=n_iter))
    lo, hi = shard_bounds(B, world, rank)
    elapsed, stats = rank_stats(dist, dev, elapsed, (hi - lo) * n_iter * args.steps)
    assert out.shape == (B, 1, H, W, C)
    roof, kern = dominant_roofline(sharded.rec._handle, lambda: sharded(frames, n_iter=n_iter), rank == 0)
    if rank == 0:
        emit({
            "roofline": roof, "kernels": kern,
            "metric": "ADMM frame-iterations/sec, batch of 64 frames 270x480x3, 20 iters (BASELINE config 4)",
            "value": round(B * n_iter * args.steps / elapsed, 1), "unit": "frame-iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4: 64 frames 270x480x3, ADMM-TV 20 iterations, frames block-sharded over the "
                                   "ranks, one all-gather per step (solver built once, outside the timed region)",
                       "frames_per_gpu": -(-B // world)},
            "all_gather_ms": sharded.gather_ms(), "all_gather_MB_per_rank": round(-(-B // world) * H * W * C * 4 / 1e6, 2),
            "engine_plan": sharded.rec._handle.plan_info(), "backend": RT.name, **stats,
        })
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def run_c5_planes(args, rank, world, dev, dist):
    """BASELINE config 5 as ONE frame over the node: the 16 depth planes x 3 channels of the stack are 48 independent
    single-plane ADMM problems, block-sharded over the ranks by
    PlaneShardedReconstructor; one all-gather of the finished planes per step.  Strong scaling."""
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd.dist import PlaneShardedReconstructor, shard_bounds

    D, H, W, C, n_iter = 16, 1080, 1920, 3, 50
    if args.test_shape:
        _, D, H, W, n_iter = args.test_shape
    g = torch.Generator(device=dev).manual_seed(0)
    psf = torch.rand((D, H, W, C), device=dev, generator=g) ** 12
    psf /= psf.norm()
    y = torch.rand((H, W, C), device=dev, generator=g)
    sharded = PlaneShardedReconstructor(lpa.ADMM, psf)
    elapsed, out = timed_steps(args, dist, lambda: sharded(y, n_iter=n_iter))
    lo, hi = shard_bounds(D * C, world, rank)
    elapsed, stats = rank_stats(dist, dev, elapsed, (hi - lo) * n_iter * args.steps)
    assert out.shape == (D, H, W, C)
    first = next(iter(sharded._solvers.values()), None)          # (a rank without units has no solver: rank 0 always has)
    if first is not None:
        roof, kern = dominant_roofline(first._handle, lambda: sharded(y, n_iter=n_iter), rank == 0)
    else:
        sharded(y, n_iter=n_iter)
        roof = kern = None
    if rank == 0:
        emit({
            "roofline": roof, "kernels": kern,
            "metric": "ADMM iterations/sec, one 1080x1920x3 frame against 16 depth planes, 50 iters (BASELINE config 5), "
                      "its 48 planes sharded over the GPUs",
            "value": round(n_iter * args.steps / elapsed, 2), "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5: 16 depth planes x 1080x1920x3, ADMM-TV 50 iterations, 48 (plane, channel) units "
                                   "block-sharded over the ranks, one all-gather per step",
                       "units_per_gpu": -(-D * C // world)},
            "plane_units_per_s_per_rank": [stats["per_rank_units_per_s_min"], stats["per_rank_units_per_s_max"]],
            "backend": RT.name, **stats,
        })
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(plan_info):
    """profiles/traffic.json: HBM bytes per launch of every hot-loop kernel from separate rocprofv3 --pmc passes
    (2 * FETCH_SIZE + WRITE_SIZE KiB, median over the steady-state dispatches of a 40-iteration call; PMC counters cannot
    be collected inside this run).  Used only when the counters were taken on the launch plan this run chose."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tf):
        return None
    try:
        tj = json.load(open(tf))
    except Exception:
        return None
    from lenslesspicam_amd import build as _build

    fp = _build.fingerprint()
    key = plan_info.split("plan module ", 1)[1].strip() if "plan module " in plan_info else None
    for entry in tj.get("plans", []):
        # the WHOLE module key (one key may be a prefix of another), and counters are only as good as the kernels they were
        # taken on: an entry carries the fingerprint of the sources it was measured with (tools/summarize_prof.py stamps
        # it); another fingerprint = stale = not reported
        if key and entry.get("plan_module") == key and entry.get("source_fingerprint") == fp:
            return entry
    return None


def kernel_table(handle, prof, traffic=None):
    """per-kernel mean launch time (HIP events on the solver's stream), algorithmic GB/s and -- where PMC counters of
    this launch plan are on file -- HBM traffic per launch and its ratio to the algorithmic bytes"""
    from lenslesspicam_amd import _native

    kernels = {}
    for i, name in enumerate(_native.KERNEL_NAMES):
        ms, n = prof[name]
        if n:
            b = handle.kernel_bytes(i)
            kernels[name] = {"ms": round(ms, 4), "launches": n, "alg_GB": round(b / 1e9, 3),
                             "GBps": round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                             "frac_of_peak": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None}
            t = (traffic or {}).get("kernels", {}).get(name)
            # (one plan module serves every batch size of a frame shape: counters taken on another batch size are not
            # this workload's -- a ratio outside [0.5, 2] can only be that)
            if t and b > 0 and 0.5 <= t["hbm_bytes_per_launch"] / b <= 2.0:
                kernels[name].update(traffic_GB=round(t["hbm_bytes_per_launch"] / 1e9, 3),
                                     traffic_over_alg=round(t["hbm_bytes_per_launch"] / b, 3), pmc_kernel=t["kernel"])
    return kernels


def dominant_roofline(handle, call, emit_it):
    """`roofline` of a sharded config's line: one more step after the timed region (every rank takes part: the step holds
    the collective) with this rank's launches bracketed; the kernel with the largest share of the step."""
    handle.profile_enable(True)
    call()
    prof = handle.profile_read()
    handle.profile_enable(False)
    if not emit_it:
        return None, None
    kern = kernel_table(handle, prof, load_traffic(handle.plan_info()))
    if not kern:        # (the SIMT emulator has no event timer: the line keeps its shape)
        return {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0,
                "traffic": None, "note": "no launch was timed on this backend"}, kern
    name = max(kern, key=lambda k: kern[k]["ms"] * kern[k]["launches"])
    v = kern[name]
    return {"bound": "hbm", "kernel": name, "achieved": v["GBps"] or 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": v["frac_of_peak"] or 0.0, "alg_bytes_per_launch": round(v["alg_GB"] * 1e9), "avg_launch_ms": v["ms"],
            "launches_timed": v["launches"], "traffic": round(v["traffic_GB"] * 1e9) if "traffic_GB" in v else None,
            "note": "rank 0's shard, one extra step after the timed region with every launch bracketed by HIP events"}, kern


def timed_config(name, rec, call, units_per_call, unit, reps, note, groups=3):
    """One BASELINE config other than the headline: `groups` samples of `reps` timed calls each after one warm-up call
    (no events inside the timed calls: on a 0.3-ms call their recording shows); `value` is the median sample, `samples`
    lists them all.  Then one more call with the kernel events on."""
    call()
    torch.cuda.synchronize()
    dts = []
    for _ in range(groups):
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / reps)
    dt = sorted(dts)[len(dts) // 2]
    rec._handle.profile_enable(True)
    call()
    prof = rec._handle.profile_read()
    rec._handle.profile_enable(False)
    kern = kernel_table(rec._handle, prof, load_traffic(rec._handle.plan_info()))
    alg = sum(v["alg_GB"] * v["launches"] for v in kern.values())               # algorithmic GB per call
    busy = sum(v["ms"] * v["launches"] for v in kern.values())                   # kernel ms per call
    return {"config": name, "engine_plan": rec._handle.plan_info(),
            "value": round(units_per_call / dt, 2), "unit": unit, "ms_per_call": round(dt * 1e3, 3),
            "samples": [round(units_per_call / v, 2) for v in dts],
            "alg_GB_per_call": round(alg, 3),
            "whole_call_frac_of_peak": round(alg / dt / HBM_PEAK_GBS, 4),       # algorithmic bytes / wall time / 8 TB/s
            "kernel_ms_per_call": round(busy, 3), "kernels": kern, "note": note}


def other_configs(dev):
    """Short legs for the BASELINE configs bench.py's headline does not cover (C1, C3, C4, C5), one GPU, synthetic
    inputs of SURVEY 8(d)'s shapes (random frames: timing does not depend on the pixel values)."""
    import lenslesspicam_amd as lpa

    def rand_inputs(D, H, W, C, B, seed=0):
        g = torch.Generator(device=dev).manual_seed(seed)
        psf = torch.rand((D, H, W, C), device=dev, generator=g) ** 12
        psf /= psf.norm()
        return psf, torch.rand((B, H, W, C), device=dev, generator=g)

    out = []
    psf, y = rand_inputs(1, 270, 480, 3, 1)
    rec = lpa.ADMM(psf)
    rec.set_data(y[0])
    out.append(timed_config("C1: 270x480x3 ADMM 5 iterations (apply = reset + 5 it + read-out)", rec,
                            lambda: rec.apply(n_iter=5, disp_iter=None), 5, "iterations/s", 20, "profile/admm.py size"))
    del rec
    psf, y = rand_inputs(1, 3040, 4056, 3, 1)
    fis = lpa.FISTA(psf)
    fis.set_data(y[0])
    out.append(timed_config("C3: 3040x4056x3 FISTA, 300 iterations (one apply)", fis,
                            lambda: fis.apply(n_iter=300, disp_iter=None), 300, "iterations/s", 1,
                            "BASELINE config 3 at its own iteration count"))
    del fis
    torch.cuda.empty_cache()
    r64 = lpa.ADMM(psf.double(), dtype="float64")
    r64.set_data(y[0].double())
    out.append(timed_config("C2 in float64 (dtype='float64', lensless/utils/io.py:645-674): 3040x4056x3 ADMM, 40 iterations",
                            r64, lambda: r64.apply(n_iter=40, disp_iter=None), 40, "iterations/s", 1,
                            f"liblpc_f64.so, same design on twice the bytes; {r64._handle.workspace_bytes() / 1e9:.1f} GB of HBM"))
    del r64
    torch.cuda.empty_cache()
    psf, y = rand_inputs(1, 270, 480, 3, 64)
    r4 = lpa.ADMM(psf)
    r4.set_data(y[:, None])
    out.append(timed_config("C4: batch of 64 x 270x480x3, ADMM 20 iterations, ONE GPU (8 GPUs: 8 frames each)", r4,
                            lambda: r4.apply_batch(n_iter=20), 64 * 20, "frame-iterations/s", 3,
                            "solver built once; sharded form: bench.py --config c4"))
    del r4
    r48 = lpa.ADMM(psf)
    r48.set_data(y[:8, None])
    out.append(timed_config("C4 shard: 8 of the 64 frames (what ONE of 8 GPUs runs), ADMM 20 iterations", r48,
                            lambda: r48.apply_batch(n_iter=20), 8 * 20, "frame-iterations/s", 10,
                            "x 8 = the C4 rate of an 8-GPU node before its one all-gather"))
    del r48
    # two frame shapes that are on nobody's list (RPi-HQ at downsample 2 and 8, lensless/hardware/sensor.py:76): their
    # compile-time-plan kernels are compiled on first use (plan modules) -- same protocol as their neighbours C2 / C1
    psf, y = rand_inputs(1, 1520, 2028, 3, 1)
    t0 = time.perf_counter()
    r6 = lpa.ADMM(psf)
    t_create = time.perf_counter() - t0
    r6.set_data(y[0])
    out.append(timed_config("off-list 1520x2028x3 (RPi-HQ, downsample 2): ADMM 100 iterations", r6,
                            lambda: r6.apply(n_iter=100, disp_iter=None), 100, "iterations/s", 2,
                            f"solver construction incl. finding / compiling the plan module: {t_create:.2f} s"))
    del r6
    psf, y = rand_inputs(1, 380, 507, 3, 1)
    t0 = time.perf_counter()
    r7 = lpa.ADMM(psf)
    t_create = time.perf_counter() - t0
    r7.set_data(y[0])
    out.append(timed_config("off-list 380x507x3 (RPi-HQ, downsample 8): ADMM 5 iterations (apply = reset + 5 it + read-out)",
                            r7, lambda: r7.apply(n_iter=5, disp_iter=None), 5, "iterations/s", 20,
                            f"solver construction incl. finding / compiling the plan module: {t_create:.2f} s"))
    del r7
    torch.cuda.empty_cache()
    psf, y = rand_inputs(16, 1080, 1920, 3, 1)
    r5 = lpa.ADMM(psf)
    r5.set_data(y[0])
    out.append(timed_config("C5: 16 depth planes x 1080x1920x3, ADMM 50 iterations", r5,
                            lambda: r5.apply(n_iter=50, disp_iter=None), 50, "iterations/s", 1,
                            f"{r5._handle.workspace_bytes() / 1e9:.1f} GB of HBM"))
    del r5
    torch.cuda.empty_cache()
    r52 = lpa.ADMM(psf[:2].contiguous())
    r52.set_data(y[0])
    out.append(timed_config("C5 per-rank share of `--config c5-planes` on 8 GPUs: 6 of the 48 (plane, channel) units = 2 "
                            "depth planes x 3 channels in ONE handle, ADMM 50 iterations", r52,
                            lambda: r52.apply(n_iter=50, disp_iter=None), 50, "iterations/s", 3,
                            "strong scaling of ONE frame's depth stack: this rate is the node's C5 rate before its all-gather"))
    del r52
    torch.cuda.empty_cache()
    return out


_JSON_OUT = None


def emit(obj):
    """The ONE JSON line of the contract, on the process's real stdout."""
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    print(json.dumps(obj), file=out, flush=True)


def main():
    global _JSON_OUT
    # stdout must carry exactly one line.  RCCL prints a start-up banner ("RCCL version : ...", 5 lines) from C on file
    # descriptor 1 when the process group initialises: keep the real stdout aside for the JSON line and point
    # descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = RT.device(local)
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:  # launched by torchrun (any world size)
        import torch.distributed as dist

        RT.init_group(dist, dev)
    assert world == args.gpus or world == 1, "launch with torchrun --nproc-per-node == --gpus"

    import lenslesspicam_amd as lpa
    from lenslesspicam_amd import _native

    if args.config == "c4":
        return run_c4(args, rank, world, dev, dist)
    if args.config == "c5-planes":
        return run_c5_planes(args, rank, world, dev, dist)

    H, W, C, n_iter = args.height, args.width, 3, args.n_iter
    log("generating synthetic inputs")
    psf, scene, y = synth_inputs(H, W, C, rank, dev)
    if args.dtype == "float64":
        psf, y = psf.double(), y.double()
    RT.sync()
    log("inputs ready; building solver")
    if args.algo == "admm":
        rec = lpa.ADMM(psf, dtype=args.dtype, n_iter=n_iter)
    else:
        rec = lpa.FISTA(psf, dtype=args.dtype, n_iter=n_iter)
    rec.set_data(y)
    hp, wp = rec._padded_shape[1], rec._padded_shape[2]

    # the single collective of the path: every rank ends a step holding all frames' results.  One flat receive buffer,
    # issued asynchronously (RCCL's own stream) so that the gather of step k rides over xGMI while step k + 1 computes;
    # the last one is waited for INSIDE the timed region.
    # (concatenated along the leading axis of one rank's result, (D = 1, H, W, C): the form every backend accepts)
    gathered = torch.empty((world * 1, H, W, C), dtype=psf.dtype, device=dev) if dist else None
    inflight = {"work": None, "send": None}

    def wait_gather():
        if inflight["work"] is not None:
            inflight["work"].wait()
            inflight["work"] = None

    def step():
        out = rec.apply(n_iter=n_iter, disp_iter=None, plot=False)
        if dist:
            wait_gather()                              # the receive buffer is free again
            inflight["send"] = out.contiguous()        # kept alive until the collective has read it
            inflight["work"] = dist.all_gather_into_tensor(gathered, inflight["send"], async_op=True)
        return out

    RT.sync()
    log(f"solver ready ({rec._handle.workspace_bytes() / 1e9:.1f} GB HBM); warm-up")
    for _ in range(args.warmup):
        step()
    wait_gather()
    RT.sync()
    log("timed region")
    # HIP events inside the timed region only around the kernel the roofline reports: bracketing all six to eight launches
    # of an iteration costs the timed rate 1.4 % (ADMM) to 3.4 % (FISTA) (tools/probe/event_overhead.py); the other kernels'
    # rows of `kernels` come from one more step after the clock has stopped
    rec._handle.profile_enable(True, kernels=["spatial", "row_fwd"] if args.algo == "admm" else ["spatial"])
    if dist:
        dist.barrier()
    RT.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    wait_gather()
    RT.sync()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, rstats = rank_stats(dist, dev, elapsed, args.steps * n_iter)
    gather_ms = None
    if dist:
        assert torch.equal(gathered[rank:rank + 1], out)      # this rank's slot of the last gather is its own result
        # the collective on its own, outside the timed region (inside it rides under the next step's iterations)
        send = out.contiguous()
        dist.all_gather_into_tensor(gathered, send)
        gather_ms = RT.timed_ms(lambda: dist.all_gather_into_tensor(gathered, send), 3)
    prof_live = rec._handle.profile_read()
    log(f"timed region done: {elapsed:.3f} s for {args.steps} step(s)")
    rec._handle.profile_enable(True)
    step()
    wait_gather()
    prof = rec._handle.profile_read()
    rec._handle.profile_enable(False)
    for k in ("spatial", "row_fwd"):                  # the roofline kernel(s): the timed region's own launches
        if prof_live.get(k, (0, 0))[1]:
            prof[k] = prof_live[k]

    # achievable-HBM yardstick measured in the same run: plain device-to-device copy
    copy_gbps = None
    if rank == 0:
        src_buf = torch.empty((1 if RT.emu else 256) * 1024 * 1024, dtype=torch.float32, device=dev)   # 1 GiB
        dst_buf = torch.empty_like(src_buf)
        dst_buf.copy_(src_buf)
        RT.sync()
        copy_gbps = 2 * src_buf.numel() * 4 / (RT.timed_ms(lambda: dst_buf.copy_(src_buf), 10) * 1e-3) / 1e9
        del src_buf, dst_buf

    result = None
    if rank == 0:
        total_iters = world * args.steps * n_iter
        value = total_iters / elapsed
        kid = _native.K_SPATIAL
        kbytes = rec._handle.kernel_bytes(kid)
        k_ms, k_n = prof["spatial"]
        achieved = kbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        tr = load_traffic(rec._handle.plan_info()) if args.algo == "admm" else None
        traffic, traffic_src = None, None
        if tr and "spatial" in tr.get("kernels", {}):
            traffic = tr["kernels"]["spatial"]["hbm_bytes_per_launch"]
            traffic_src = (f"profiles/traffic.json: separate rocprofv3 --pmc passes of {tr['kernels']['spatial']['kernel']} "
                           f"(snapshot {tr.get('snapshot', '?')}, median over the steady-state dispatches of a "
                           f"{tr.get('n_iter', '?')}-iteration call; PMC counters cannot be collected inside this run)")
        kernels = kernel_table(rec._handle, prof, tr)
        # SURVEY section 8(d) prices "the fused prox / update kernel" at 15R + R0: everything of an iteration that lives in
        # the image domain.  In this engine that work is TWO launches -- the tiled TV / W kernel (roofline.achieved above)
        # and the X half inside the forward row kernel -- so the scope-equivalent figure is their bytes over their time
        combined = None
        if args.algo == "admm" and "spatial" in kernels and "row_fwd" in kernels and kernels["row_fwd"]["ms"]:
            cb = (kernels["spatial"]["alg_GB"] + kernels["row_fwd"]["alg_GB"])
            cm = kernels["spatial"]["ms"] + kernels["row_fwd"]["ms"]
            combined = {"scope": "SURVEY 8(d)'s whole prox / update kernel = tiled TV / W kernel + forward rows with the X half; "
                                 "both timed by HIP events inside the timed region",
                        "bytes": round(cb * 1e9), "ms": round(cm, 4), "achieved": round(cb / (cm * 1e-3), 1),
                        "frac": round(cb / (cm * 1e-3) / HBM_PEAK_GBS, 4)}
        moved_gb = sum(v["alg_GB"] for v in kernels.values())
        result = {
            "metric": "ADMM iterations/sec at 4056x3040x3, 100 iters" if args.algo == "admm"
            else "FISTA iterations/sec at 4056x3040x3",
            "value": round(value, 3),
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "float32" else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"C2: one {H}x{W}x{C} frame per GPU, {args.algo.upper()}"
                            f"{'-TV' if args.algo == 'admm' else ''} {n_iter} iterations per step",
                "frame": [H, W, C], "padded_fft_frame": [hp, wp], "n_iter": n_iter,
                "frames_per_gpu": 1, "parallelism": f"frames sharded over {world} GPU(s), one all-gather per step",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("LPC_K_SPATIAL = ADMM prox / dual-update kernel; split of the image-domain work: "
                           + next((p.strip() for p in rec._handle.plan_info().split(";") if "image-domain" in p or "TV / W" in p),
                                  "")) if args.algo == "admm"
                else "k_rinv_gd_update (inverse rows + fused projected update)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "alg_bytes_per_launch": kbytes, "avg_launch_ms": round(k_ms, 4), "launches_timed": k_n,
                "traffic": traffic, "traffic_source": traffic_src,
                "prox_update_scope": combined,
            },
            "kernels": kernels,
            "kernels_note": "spatial, row_fwd: HIP events inside the timed region (the roofline scope); the other rows: one more "
                            "step after the timed region with every launch bracketed",
            "alg_GB_per_iteration": round(sum(v["alg_GB"] for v in kernels.values()), 3),
            "survey_model_GB_per_iteration": round(rec._handle.model_bytes() / 1e9, 3),
            # two different statements: the first prices the run on SURVEY 8(d)'s MODEL of an iteration (19R + R0 + 13.5S:
            # "fraction of the model roofline" -- it credits bytes the engine no longer moves); the second is HBM
            # utilisation, the bytes the engine's kernels do move per iteration over the same wall time
            "whole_iteration_frac_of_peak": {
                "on_survey_model_bytes": round(rec._handle.model_bytes() * total_iters / world / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                "on_bytes_moved": round(moved_gb * total_iters / world / elapsed / HBM_PEAK_GBS, 4),
                "survey_model_GB": round(rec._handle.model_bytes() / 1e9, 3), "bytes_moved_GB": round(moved_gb, 3)},
            "backend": RT.name,
            "device_copy_GBps": round(copy_gbps, 1) if copy_gbps else None,
            "hbm_workspace_GB": round(rec._handle.workspace_bytes() / 1e9, 2),
            "engine_plan": rec._handle.plan_info(),
            "all_gather_ms": round(gather_ms, 3) if gather_ms is not None else None,
            "all_gather_MB_per_rank": round(H * W * C * (8 if args.dtype == "float64" else 4) / 1e6, 2),
            **rstats,
        }

    if rank == 0 and not args.no_parity and args.dtype == "float32":
        par = reference_parity(out, scene, H, W, n_iter, args.algo)
        if par:
            result["parity"] = par
            log(f"parity vs the reference's samples: {par['engine_f32_vs_reference_f64']:.2e} (the reference's own float32: "
                f"{par['reference_f32_vs_reference_f64']:.2e}), PSNR delta {par['psnr_delta_db']:+.2e} dB")
            assert abs(par["psnr_delta_db"]) <= 0.01, par

    # ---- CPU baseline: rank 0, N == 1 only -------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.algo == "admm" and args.dtype == "float32":
        import psutil

        from oracle import lensless_oracle as orc      # the CPU restatement: the baseline leg only

        cores = os.cpu_count() or 1
        avail_gb = psutil.virtual_memory().available / 1e9
        need_gb = 32.0 * hp * wp * C * 4 / 1e9  # ~32 padded float32 arrays incl. complex spectra + temporaries
        bH, bW, note = H, W, ""
        if avail_gb < need_gb * 1.3:
            bH, bW = 1520, 2028
            note = f" (host has {avail_gb:.0f} GB free < {need_gb * 1.3:.0f} GB: sampled at {bH}x{bW} instead)"
        if (bH, bW) == (H, W):
            psf_c, y_c = psf.cpu().numpy(), y.cpu().numpy()
        else:
            psf_c = orc.synthetic_psf(1, bH, bW, C, seed=0)
            y_c = np.random.default_rng(0).random((bH, bW, C), dtype=np.float32)
        torch.set_num_threads(min(cores, 64))
        log(f"cpu baseline: oracle set-up at {bH}x{bW}")
        o = orc.ADMMOracle(psf_c)
        o.set_data(y_c)
        # thread count: every host thread over-subscribes the FFTs of a 12-MP frame (round 1: 256 threads were slower
        # than 8) -- one iteration each at 64 threads and at all of them decides; both count as iterations 1 and 2
        trial = {}
        for nt in sorted({min(cores, 64), cores}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            o.step()
            trial[nt] = time.perf_counter() - t0
        best = min(trial, key=trial.get)
        torch.set_num_threads(best)
        log(f"cpu baseline: 1 iteration took {', '.join(f'{v:.1f} s at {k} threads' for k, v in trial.items())}; "
            f"timing {args.cpu_iters} iterations at {best}")
        t0 = time.perf_counter()
        for _ in range(args.cpu_iters):
            o.step()
        cpu_s = time.perf_counter() - t0
        cpu_ips = args.cpu_iters / cpu_s
        cpu_done = len(trial) + args.cpu_iters
        log(f"cpu baseline: {cpu_s:.1f} s for {args.cpu_iters} iterations")
        result["cpu_baseline"] = {
            "value": round(cpu_ips, 5), "unit": "iterations/s", "cores": best, "host_threads": cores,
            "kind": "port",
            "sample": f"{args.cpu_iters} ADMM iterations (the 3rd..{cpu_done}th) of the same {bH}x{bW}x{C} frame; oracle = "
                      f"torch-CPU float32 restatement of the reference, set-up excluded; thread count = the faster of "
                      f"{sorted(trial)} on a 1-iteration trial{note}",
        }
        result["speedup_vs_cpu"] = round(result["value"] / cpu_ips, 1) if (bH, bW) == (H, W) else None
        del o

    if rank == 0 and world == 1 and not args.no_other_configs and args.algo == "admm" and args.dtype == "float32":
        log("other BASELINE configs (C1, C3, C4, C5)")
        del rec
        torch.cuda.empty_cache()
        result["other_configs"] = other_configs(dev)
        log("other configs done")

    if rank == 0:
        emit(result)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
