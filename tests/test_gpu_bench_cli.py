"""bench.py as the driver launches it: the single-GPU command line, and the torch.distributed.run form with one rank
forced through the row-sharded path (RCCL collectives inside the captured step -- the closest a one-GPU box gets to the
N > 1 launch).  Both must exit 0 and print ONE JSON line carrying the contract's keys."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd, env):
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_single_gpu_command_line():
    env = dict(os.environ)
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], env)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 10 and d["config"]["checked"] is True
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["launch"].startswith("hipGraph") and d["ms_per_step"] < 1.0
    assert "train_plus_eval_ms_per_epoch" in d["config"] and "literal_ms_per_step" in d["config"]


def test_torchrun_launch_with_collectives_in_the_captured_step():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "10", "--warmup", "3",
              "--no-cpu-baseline", "--force-sharded"], env)
    assert KEYS <= set(d) and d["config"]["checked"] is True
    assert d["config"]["launch"].startswith("hipGraph"), d["config"]["launch"]
    assert d["config"]["shard"]["work_max_over_mean"] <= 1.05


def test_bench_starts_its_own_ranks():
    """``python bench.py --gpus N`` without a launcher around it starts one rank per GPU itself
    (bench.self_launch: torch.distributed.run, 127.0.0.1 rendezvous).  On the one-GPU box the same entry is reached with
    ``--gpus 1 --force-sharded``: RCCL initialised, the row-sharded step with its collectives captured, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--force-sharded"], env)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["config"]["checked"] is True
    assert d["config"]["launch"].startswith("hipGraph"), d["config"]["launch"]
    assert d["config"]["shard"]["work_max_over_mean"] <= 1.05
