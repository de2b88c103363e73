"""Everything about 8 GPUs that does not need 8 GPUs: world size 8 on CPU (gloo), every rank on the SIMT-emulator build of
the kernels -- the sharding logic, the collective and bench.py's N > 1 branch are the product code.

  * BASELINE config 4's split (64 frames = 8 x 8, `ShardedReconstructor`), an uneven batch (13 = 2 2 2 2 2 1 1 1) and
    config 5's plane split (16 depth planes x 3 channels = 48 units, 6 per rank, `PlaneShardedReconstructor`): every rank
    ends with the whole result, bit for bit the un-sharded solver's (reference semantics: a batch equals its single frames,
    /root/reference/test/test_algos.py:198-229; depth planes are independent, test/test_convolver.py:32-49);
  * eight ranks that find the plan module of a new frame shape missing at the same moment (one process per GPU on a fresh
    node): one module file, no temporaries left, everybody on the module;
  * `bench.py --gpus 8` the way the driver launches its scaling runs, headline, `--config c4` and `--config c5-planes`:
    the JSON line's fields (rccl_world 8, per-rank rates, shard sizes, all_gather_MB_per_rank, roofline).

No 8-GPU node was ever available to this build (SCALE_r0x.json: skipped): no scaling CURVE has been measured; this file is
what can be proven without one.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port, emu_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPC_EMU_THREADS="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lenslesspicam_amd import _native, recon

    lib = _native.Lib(emu_path)
    recon.runtime = lambda dtype="float32": (lib, torch.device("cpu"))


def _worker(rank, world, port, emu_path, out_dir):
    _setup(rank, world, port, emu_path)
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd.dist import PlaneShardedReconstructor, ShardedReconstructor, shard_bounds

    rng = np.random.default_rng(8)
    psf = rng.random((1, 12, 16, 3), dtype=np.float32) ** 4
    psf /= np.linalg.norm(psf.ravel())
    frames = rng.random((64, 12, 16, 3), dtype=np.float32)
    kw = dict(tau=2e-6, mu2=1e-4)
    sr = ShardedReconstructor(lpa.ADMM, psf, **kw)
    # C4's split: 64 frames, 8 per rank
    assert shard_bounds(64, world, rank) == (8 * rank, 8 * rank + 8)
    full = sr(frames, n_iter=3)
    assert full.shape == (64, 1, 12, 16, 3)
    whole = lpa.ADMM(psf, **kw)
    whole.set_data(frames[:, None])
    assert np.array_equal(full, whole.apply_batch(n_iter=3)), rank                # == the un-sharded batch, bit for bit
    lo, hi = shard_bounds(64, world, rank)
    for b in (lo, hi - 1):                                                        # ... == single frames
        single = lpa.ADMM(psf, **kw)
        single.set_data(frames[b])
        assert np.array_equal(full[b], single.apply(n_iter=3, disp_iter=None)), (rank, b)
    # uneven: 13 frames over 8 ranks (the solver is re-used at another shard size), and fewer frames than ranks
    for nb in (13, 5):
        got = sr(frames[:nb], n_iter=2)
        whole.set_data(frames[:nb, None])
        assert got.shape == (nb, 1, 12, 16, 3) and np.array_equal(got, whole.apply_batch(n_iter=2)), (rank, nb)
    # C5's split: 16 depth planes x 3 channels = 48 (plane, channel) units, 6 per rank = two whole RGB planes
    psf16 = rng.random((16, 12, 16, 3), dtype=np.float32) ** 4
    y = rng.random((12, 16, 3), dtype=np.float32)
    ps = PlaneShardedReconstructor(lpa.ADMM, psf16, **kw)
    got = ps(y, n_iter=3)
    assert len(ps.units) == 48 and list(ps._solvers) == [((2 * rank, 2 * rank + 1), (0, 1, 2))], list(ps._solvers)
    stack = lpa.ADMM(psf16, **kw)
    stack.set_data(y)
    assert got.shape == (16, 12, 16, 3) and np.array_equal(got, stack.apply(n_iter=3, disp_iter=None, plot=False)), rank
    np.save(os.path.join(out_dir, f"w8_rank{rank}.npy"), np.concatenate([full.ravel(), got.ravel()]))
    dist.barrier()
    dist.destroy_process_group()


def test_c4_and_c5_splits_world8_gloo(emu_lib, tmp_path):
    mp.spawn(_worker, args=(WORLD, _port(), emu_lib.path, str(tmp_path)), nprocs=WORLD, join=True)
    ref = np.load(tmp_path / "w8_rank0.npy")
    for r in range(1, WORLD):
        assert np.array_equal(ref, np.load(tmp_path / f"w8_rank{r}.npy")), r      # every rank holds the same whole result


def _jit_race_worker(rank, world, port, emu_path, out_dir):
    _setup(rank, world, port, emu_path)
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd.dist import reconstruct_sharded

    rng = np.random.default_rng(3)
    psf = rng.random((1, 20, 36, 1), dtype=np.float32) ** 4
    frames = rng.random((8, 20, 36, 1), dtype=np.float32)
    opts = {"jit_min_points": 0, "module_dir": os.path.join(out_dir, "modules")}      # an EMPTY module directory
    dist.barrier()                                               # all eight ranks reach lpc_create together ...
    full = reconstruct_sharded(lpa.ADMM, psf, frames, n_iter=3, tau=2e-6, mu2=1e-4, engine_options=opts)
    probe = lpa.ADMM(psf, tau=2e-6, mu2=1e-4, engine_options=opts)
    assert "plan module" in probe._handle.plan_info()            # ... and all ended up on the module, not the fallback
    np.save(os.path.join(out_dir, f"jit8_rank{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_compile_the_same_plan_module_at_once(emu_lib, tmp_path):
    mp.spawn(_jit_race_worker, args=(WORLD, _port(), emu_lib.path, str(tmp_path)), nprocs=WORLD, join=True)
    ref = np.load(tmp_path / "jit8_rank0.npy")
    for r in range(1, WORLD):
        assert np.array_equal(ref, np.load(tmp_path / f"jit8_rank{r}.npy")), r
    files = os.listdir(tmp_path / "modules")
    assert len([f for f in files if f.endswith(".so")]) == 1 and not [f for f in files if ".tmp" in f], files


# ------------------------------------------------------------------------- bench.py --gpus 8 under torch.distributed.run --
def _bench(extra):
    env = dict(os.environ, LPC_BENCH_BACKEND="emu", LPC_EMU_THREADS="1", OMP_NUM_THREADS="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={WORLD}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(WORLD), "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout           # the contract: ONE JSON line on stdout (rank 0), nothing else
    j = json.loads(lines[0])
    assert j["n_gpus"] == WORLD and j["rccl_world"] == WORLD and len(j["per_rank_s"]) == WORLD and j["backend"] == "simt-emu"
    assert 0 < j["per_rank_units_per_s_min"] <= j["per_rank_units_per_s_max"]
    assert j["value"] > 0 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0 and j["roofline"]["frac"] >= 0       # (the emulator has no event timer)
    assert "cpu_baseline" not in j                                                # rank 0 at N == 1 only
    return j


def test_bench_headline_eight_ranks(emu_lib):
    j = _bench(["--height", "20", "--width", "24", "--n-iter", "3", "--no-other-configs"])
    assert j["scaling"] == "weak" and j["unit"] == "iterations/s" and j["config"]["frames_per_gpu"] == 1
    assert abs(j["value"] - WORLD * 2 * 3 / (j["ms_per_step"] * 2 / 1e3)) <= 1e-3 * j["value"]   # whole-job aggregate
    assert j["all_gather_ms"] is not None and j["all_gather_MB_per_rank"] == round(20 * 24 * 3 * 4 / 1e6, 2)


@pytest.mark.parametrize("config,shape,unit,per_gpu", [("c4", "64,1,12,16,3", "frame-iterations/s", ("frames_per_gpu", 8)),
                                                       ("c5-planes", "1,16,12,16,3", "iterations/s", ("units_per_gpu", 6))])
def test_bench_sharded_configs_eight_ranks(emu_lib, config, shape, unit, per_gpu):
    """--config c4 at its own split (64 frames = 8 x 8) and --config c5-planes at its own (48 units = 6 per rank)"""
    j = _bench(["--config", config, "--test-shape", shape])
    assert j["scaling"] == "strong" and j["unit"] == unit and j["config"][per_gpu[0]] == per_gpu[1]
    if config == "c4":
        assert j["all_gather_MB_per_rank"] == round(8 * 12 * 16 * 3 * 4 / 1e6, 2) and j["all_gather_ms"] >= 0
        assert abs(j["value"] - 64 * 3 * 2 / (j["ms_per_step"] * 2 / 1e3)) <= 1e-3 * j["value"]
