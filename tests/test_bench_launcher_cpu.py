"""bench.py's own launcher (bench.self_launch): ``python bench.py --gpus N`` with no torch.distributed
launcher around it starts one rank per GPU; a process that already is a rank, or the plain one-GPU call, runs the benchmark
itself.  CPU: the decision and the command line (``--dry-run-launch``), and one real launch whose ranks get as far as a box
without a GPU lets them -- the device check of bench.main()."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RANK_ENV = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")


def _dry(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in _RANK_ENV}
    env.update(extra_env or {})
    res = subprocess.run([sys.executable, "bench.py", "--dry-run-launch"] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=120)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = res.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    return json.loads(lines[0])["launch"]


def test_one_gpu_runs_in_this_process():
    assert _dry([]) is None
    assert _dry(["--gpus", "1", "--steps", "5", "--warmup", "2"]) is None


def test_a_rank_never_launches_again():
    assert _dry(["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}) is None


def test_n_gpus_start_n_ranks_with_the_same_command_line():
    for n in (2, 4, 8):
        launch = _dry(["--gpus", str(n), "--steps", "20", "--warmup", "5"])
        cmd, env = launch["cmd"], launch["env"]
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and f"--nproc-per-node={n}" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
        tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
        assert tail == ["--gpus", str(n), "--steps", "20", "--warmup", "5"]          # (the dry-run flag is not passed on)
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_force_sharded_reaches_rccl_through_the_same_entry():
    launch = _dry(["--gpus", "1", "--force-sharded"])
    assert "--nproc-per-node=1" in launch["cmd"] and launch["cmd"][-3:] == ["--gpus", "1", "--force-sharded"]


def test_a_real_launch_reaches_the_ranks():
    """No GPU here: every started rank stops at bench.main()'s device check -- which proves the ranks were started with
    RANK / WORLD_SIZE set and bench.py as their program (the launcher's exit status is the job's: non-zero)."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU box runs the real thing (tests/test_gpu_bench_cli.py)")
    env = {k: v for k, v in os.environ.items() if k not in _RANK_ENV}
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = res.stderr.decode()
    assert res.returncode != 0
    assert "bench.py needs an MI355X" in err, err[-3000:]
    assert "must be launched through" not in err
