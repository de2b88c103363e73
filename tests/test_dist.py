"""World-size-2 test of the sharded batch path on CPU (gloo).  Each rank drives the SIMT-emulator
build (the CPU suite has no GPU); the collective and the sharding logic are the product code."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, emu_path, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPC_EMU_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd import _native, recon
    from lenslesspicam_amd.dist import reconstruct_sharded, shard_bounds

    lib = _native.Lib(emu_path)
    recon.runtime = lambda dtype="float32": (lib, torch.device("cpu"))
    rng = np.random.default_rng(0)
    psf = rng.random((1, 12, 16, 3), dtype=np.float32) ** 4
    psf /= np.linalg.norm(psf.ravel())
    frames = rng.random((3, 12, 16, 3), dtype=np.float32)      # 3 frames over 2 ranks: uneven (2 + 1)
    full = reconstruct_sharded(lpa.ADMM, psf, frames, n_iter=6, tau=2e-6, mu2=1e-4)
    assert full.shape == (3, 1, 12, 16, 3)
    lo, hi = shard_bounds(3, world, rank)
    for b in range(3):                                          # every rank holds the whole batch ...
        single = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
        single.set_data(frames[b])
        ref = single.apply(n_iter=6, disp_iter=None)
        assert np.array_equal(full[b], ref), (rank, b)          # ... bit-identical to un-sharded runs
    fis = reconstruct_sharded(lpa.FISTA, torch.from_numpy(psf), torch.from_numpy(frames), n_iter=4)
    assert isinstance(fis, torch.Tensor) and fis.shape == (3, 1, 12, 16, 3)
    # one solver kept alive over batches of different sizes: even (4 = 2 + 2), uneven (5 = 3 + 2) and a batch
    # smaller than the world (1 frame: rank 1 has an EMPTY shard and only takes part in the gather)
    from lenslesspicam_amd.dist import ShardedReconstructor

    sr = ShardedReconstructor(lpa.ADMM, psf, tau=2e-6, mu2=1e-4)
    more = rng.random((5, 12, 16, 3), dtype=np.float32)
    for nb in (4, 5, 1, 5):
        got = sr(more[:nb], n_iter=3)
        assert got.shape == (nb, 1, 12, 16, 3)
        for b in range(nb):
            single = lpa.ADMM(psf, tau=2e-6, mu2=1e-4)
            single.set_data(more[b])
            assert np.array_equal(got[b], single.apply(n_iter=3, disp_iter=None)), (rank, nb, b)
    # lifetime of a result: by default every call hands out a tensor the caller owns -- collecting the results
    # of consecutive calls is safe; reuse_output=True is the documented zero-copy ring (valid until the call after the next)
    tp, tf = torch.from_numpy(psf), torch.from_numpy(more[:4])
    own = ShardedReconstructor(lpa.ADMM, tp, tau=2e-6, mu2=1e-4)
    outs = [own(tf * s, n_iter=2) for s in (1.0, 0.5, 0.25)]
    assert len({o.data_ptr() for o in outs}) == 3
    for o, s in zip(outs, (1.0, 0.5, 0.25)):
        assert torch.equal(o, own(tf * s, n_iter=2))
    ring = ShardedReconstructor(lpa.ADMM, tp, reuse_output=True, tau=2e-6, mu2=1e-4)
    r0 = ring(tf, n_iter=2)
    keep = r0.clone()
    r1 = ring(tf * 0.5, n_iter=2)
    assert torch.equal(r0, keep) and r1.data_ptr() != r0.data_ptr()      # still valid after ONE more call ...
    r2 = ring(tf * 0.25, n_iter=2)
    assert r2.data_ptr() == r0.data_ptr() and not torch.equal(r0, keep)  # ... and overwritten by the one after that
    reused = reconstruct_sharded(lpa.ADMM, psf, frames, n_iter=6, solver=sr.rec)      # solver= skips construction
    assert np.array_equal(reused, full)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_world2_gloo(emu_lib, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, emu_lib.path, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)


def _plane_worker(rank, world, port, emu_path, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPC_EMU_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd import _native, recon
    from lenslesspicam_amd.dist import PlaneShardedReconstructor

    lib = _native.Lib(emu_path)
    recon.runtime = lambda dtype="float32": (lib, torch.device("cpu"))
    rng = np.random.default_rng(1)
    psf = rng.random((2, 12, 16, 3), dtype=np.float32) ** 4      # 2 depth planes x 3 channels
    psf /= np.linalg.norm(psf.ravel())
    y = rng.random((12, 16, 3), dtype=np.float32)
    # ADMM: 6 (depth, channel) units over the ranks; FISTA / Nesterov: 3 channel units (step size and start value
    # are taken per channel ACROSS depth, gd.py:100-112)
    for cls, kw, n_units in ((lpa.ADMM, dict(tau=2e-6, mu2=1e-4), 6), (lpa.FISTA, {}, 3),
                             (lpa.NesterovGradientDescent, {}, 3)):
        sharded = PlaneShardedReconstructor(cls, psf, **kw)
        assert len(sharded.units) == n_units
        got = sharded(y, n_iter=5)
        if cls is lpa.ADMM:       # 6 units over 2 ranks = one whole depth plane each: ONE RGB solver per rank
            assert list(sharded._solvers) == [((rank,), (0, 1, 2))], list(sharded._solvers)
        whole = cls(psf, **kw)
        whole.set_data(y)
        ref = whole.apply(n_iter=5, disp_iter=None, plot=False)
        assert got.shape == ref.shape == (2, 12, 16, 3)
        assert np.array_equal(got, ref), (rank, cls.__name__, float(np.abs(got - ref).max()))
        again = sharded(y[None] * np.float32(0.5), n_iter=2)      # the solvers are kept; (1, H, W, C) accepted
        whole.set_data(y * np.float32(0.5))
        assert np.array_equal(again, whole.apply(n_iter=2, disp_iter=None, plot=False))
    # 3 depth planes: 9 units = 5 + 4 -- neither rank holds whole planes only, so each runs one gray solver per channel
    psf3 = rng.random((3, 12, 16, 3), dtype=np.float32) ** 4
    sh3 = PlaneShardedReconstructor(lpa.ADMM, psf3, tau=2e-6, mu2=1e-4)
    got3 = sh3(y, n_iter=4)
    assert len(sh3._solvers) == 3 and all(k[1] in ((0,), (1,), (2,)) for k in sh3._solvers), list(sh3._solvers)
    whole3 = lpa.ADMM(psf3, tau=2e-6, mu2=1e-4)
    whole3.set_data(y)
    assert np.array_equal(got3, whole3.apply(n_iter=4, disp_iter=None, plot=False))
    gray = psf[:1, :, :, :1].copy()                               # one unit, two ranks: rank 1 only gathers
    one = PlaneShardedReconstructor(lpa.ADMM, torch.from_numpy(gray))(torch.from_numpy(y[:, :, :1].copy()), n_iter=3)
    solo = lpa.ADMM(torch.from_numpy(gray))
    solo.set_data(torch.from_numpy(y[:, :, :1].copy()))
    assert isinstance(one, torch.Tensor) and torch.equal(one, solo.apply(n_iter=3, disp_iter=None, plot=False))
    np.save(os.path.join(out_dir, f"plane_rank{rank}.npy"), got)
    dist.barrier()
    dist.destroy_process_group()


def test_plane_sharded_single_frame_world2_gloo(emu_lib, tmp_path):
    """SURVEY section 8e (optional row): ONE frame split by colour channel / depth plane over the ranks."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_plane_worker, args=(2, port, emu_lib.path, str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "plane_rank0.npy"), np.load(tmp_path / "plane_rank1.npy"))


def test_shard_bounds_cover_everything():
    from lenslesspicam_amd.dist import shard_bounds

    for n in (0, 1, 3, 8, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _jit_race_worker(rank, world, port, emu_path, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPC_EMU_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lenslesspicam_amd as lpa
    from lenslesspicam_amd import _native, recon
    from lenslesspicam_amd.dist import reconstruct_sharded

    lib = _native.Lib(emu_path)
    recon.runtime = lambda dtype="float32": (lib, torch.device("cpu"))
    rng = np.random.default_rng(3)
    psf = rng.random((1, 20, 44, 1), dtype=np.float32) ** 4
    frames = rng.random((2, 20, 44, 1), dtype=np.float32)
    opts = {"jit_min_points": 0, "module_dir": os.path.join(out_dir, "modules")}     # an EMPTY module directory
    dist.barrier()                                               # both ranks reach lpc_create together ...
    full = reconstruct_sharded(lpa.ADMM, psf, frames, n_iter=4, tau=2e-6, mu2=1e-4, engine_options=opts)
    probe = lpa.ADMM(psf, tau=2e-6, mu2=1e-4, engine_options=opts)
    assert "plan module" in probe._handle.plan_info()            # ... and both ended up on the module, not the fallback
    np.save(os.path.join(out_dir, f"jit_rank{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_compile_the_same_plan_module_at_once(emu_lib, tmp_path):
    """One process per GPU: every rank of a node may find the module of a new frame shape missing at the same moment.
    Each compiles into a private temporary and renames it into place (lpc_jit.cpp) -- both must come up on the module
    and agree bit for bit with a single-process run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_jit_race_worker, args=(2, port, emu_lib.path, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "jit_rank0.npy"), np.load(tmp_path / "jit_rank1.npy")
    assert np.array_equal(a, b)
    mods = [f for f in os.listdir(tmp_path / "modules") if f.endswith(".so")]
    assert len(mods) == 1 and not [f for f in os.listdir(tmp_path / "modules") if ".tmp" in f], os.listdir(tmp_path / "modules")
