"""bench.py's multi-rank plumbing without a GPU: `python bench.py --gpus 2 --dry` launches its own two ranks (torch.distributed.run,
gloo on 127.0.0.1), runs the per-step gather + barrier-bracketed timing and prints exactly one JSON line with n_gpus = 2; asking for
more GPUs than are visible fails loudly instead of silently running a smaller job."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(300)
def test_dry_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--dry", "--steps", "4", "--warmup", "1", "--pairs-per-gpu", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry"] is True and d["gathers_ok"] is True
    assert d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 4 * 2 * 2 / (d["ms_per_step"] * 4 / 1e3)) < 1e-6 * d["value"]       # whole-job pairs / max-over-ranks time


@pytest.mark.timeout(400)
def test_dry_launched_the_drivers_way_eight_ranks():
    """The driver's own command line for the scaling curve -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 8 ...`` -- on CPU / gloo: eight ranks, one line from rank 0, every step's gather complete and
    in rank order, value = whole-job pairs / max-over-ranks time."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "8", "--dry", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=360, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["dry"] is True and d["gathers_ok"] is True and d["steps"] == 3 and d["scaling"] == "weak"
    assert abs(d["value"] - 3 * 8 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_one_gpu_rank_through_rccl_gather_and_graph_replay():
    """What an N > 1 rank does, on the one GPU the pool hands out: bench.py under torch.distributed.run with world size 1 and
    S2M2_BENCH_FORCE_GATHER=1 -> RCCL communicator (with its watchdog thread) next to hipGraph capture and replay, the per-step asynchronous
    gather of the three maps, barrier-bracketed timing, one JSON line.  Small geometry: this is a plumbing test, not a measurement."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(S2M2_BENCH_FORCE_GATHER="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "6", "--warmup", "3", "--height", "256", "--width", "320",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=840, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] > 0 and d["roofline"]["launches_timed"] == 6
    assert "RCCL gather" in d["config"]["parallelism"]


@pytest.mark.timeout(120)
def test_dry_single_rank_line():
    r = _run(["--dry", "--steps", "3", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and d["steps"] == 3


@pytest.mark.timeout(120)
def test_world_size_mismatch_and_missing_gpus_fail_loudly():
    # launched "by torchrun" with a world size that contradicts --gpus
    r = _run(["--gpus", "4", "--dry"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
    # real run asking for more GPUs than are visible: refuses (exit code 2) before launching anything
    want = torch.cuda.device_count() + 1 if torch.cuda.device_count() > 0 else 2
    r = _run(["--gpus", str(want), "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "visible" in r.stderr
    assert r.stdout.strip() == ""
