"""bench.py's N > 1 branch, executed: two ranks under `torch.distributed.run`, exactly the command line the driver uses for
its scaling runs, with LPC_BENCH_BACKEND=emu (bench.py: DeviceRuntime -- CPU tensors, the SIMT-emulator build of the
kernels, a gloo group in place of RCCL).  Everything else is the product code of bench.py: step() / wait_gather() with the
asynchronous all-gather, rank_stats(), the `gathered[rank] == out` check, run_c4() on ShardedReconstructor and
run_c5_planes() on PlaneShardedReconstructor, and the ONE JSON line on the real stdout.  No 8-GPU node was ever available
to this build (SCALE_r0x.json: skipped), so this is the only place that branch runs before the driver's node runs it.
Reference semantics of the batch: /root/reference/test/test_algos.py:198-229 (a batch equals its single frames)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(emu_lib, extra, nproc=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LPC_BENCH_BACKEND="emu", LPC_EMU_THREADS="2", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout           # the contract: ONE JSON line on stdout (rank 0), nothing else
    return json.loads(lines[0])


def _common(j, world):
    assert j["n_gpus"] == world and j["steps"] == 2 and j["warmup"] == 1 and j["backend"] == "simt-emu"
    assert j["rccl_world"] == world and len(j["per_rank_s"]) == world
    assert j["per_rank_units_per_s_min"] <= j["per_rank_units_per_s_max"]
    assert j["value"] > 0 and j["ms_per_step"] > 0 and j["higher_is_better"] is True
    if "all_gather_ms" in j:
        assert j["all_gather_ms"] is not None and j["all_gather_ms"] >= 0


def test_headline_branch_two_ranks(emu_lib):
    """--config c2 (the headline workload) at a frame the emulator finishes in seconds: weak scaling, one frame per rank,
    the asynchronous all-gather of step k under step k + 1, value = iterations of all ranks / max-over-ranks time."""
    j = _run(emu_lib, ["--height", "20", "--width", "24", "--n-iter", "3", "--no-other-configs"])
    _common(j, 2)
    assert j["scaling"] == "weak" and j["unit"] == "iterations/s"
    assert j["config"]["frame"] == [20, 24, 3] and j["config"]["frames_per_gpu"] == 1
    assert abs(j["value"] - 2 * 2 * 3 / (j["ms_per_step"] * 2 / 1e3)) <= 1e-3 * j["value"]      # whole-job aggregate
    assert "cpu_baseline" not in j and "parity" not in j                                        # rank 0, N == 1 only
    assert j["roofline"]["bound"] == "hbm" and "kernels" in j


@pytest.mark.parametrize("config,shape,unit", [("c4", "5,1,12,16,3", "frame-iterations/s"), ("c5-planes", "1,2,12,16,3", "iterations/s")])
def test_sharded_configs_two_ranks(emu_lib, config, shape, unit):
    """--config c4 (batch block-sharded over the ranks: 5 frames = 3 + 2) and --config c5-planes (one frame's 6 (plane,
    channel) units over the ranks), strong scaling, one all-gather per step."""
    j = _run(emu_lib, ["--config", config, "--test-shape", shape])
    _common(j, 2)
    assert j["scaling"] == "strong" and j["unit"] == unit
