"""CPU: `python bench.py --gpus N` started WITHOUT a launcher becomes the launcher (one rank per GPU under torch.distributed.run, the
command line the driver itself uses for N > 1), and every rank of a job that finds fewer GPUs than ranks says so and exits."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_relaunch_command_is_the_drivers_command_line():
    cmd = bench.relaunch_command(["--gpus", "8", "--steps", "5", "--warmup", "2"], 8, port=29611, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"     # the container hostname may not resolve
    assert cmd[cmd.index("--master-port") + 1] == "29611"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]   # the user's flags reach every rank unchanged
    env = bench.relaunch_env({"PATH": "/bin"})
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert bench.relaunch_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # never overrides the caller


def test_plain_invocation_spawns_ranks_and_reports_missing_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    vis = torch.cuda.device_count()
    # both ranks were started (i.e. the re-exec happened) and each one explains itself
    assert "2 GPUs needed, %d visible (rank 0)" % vis in p.stderr
    assert "2 GPUs needed, %d visible (rank 1)" % vis in p.stderr
    assert p.stdout.strip() == ""          # stdout stays reserved for the ONE JSON line


def test_rehearsal_without_a_gpu_says_so_on_every_rank():
    """`--rehearsal` drops the one-GPU-per-rank requirement (all ranks share device 0) but still needs that one device."""
    import torch
    if torch.cuda.device_count() >= 1:
        import pytest
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(29900 + os.getpid() % 90)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--pool-total", "128", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "--rehearsal: needs one GPU, none visible (rank 0)" in p.stderr and "(rank 1)" in p.stderr
    assert p.stdout.strip() == ""
