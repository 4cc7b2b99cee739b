"""GPU: the driver's N = 8 command path rehearsed on the box's ONE GPU -- `python bench.py --gpus 8 --pool-total 16384 --rehearsal`
becomes the launcher (torch.distributed.run, 8 ranks, 127.0.0.1 rendezvous), every rank takes its 32-aligned block of the pool,
runs the real kernels on device 0, the per-step gather (gloo, host tensors: RCCL refuses two ranks on one device) brings the
(overlap, yaw) scores to rank 0 in pool order, every rank re-evaluates windows of its block without the Delta cache, and rank 0 alone
prints ONE JSON line -- whose gathered sweep must be `torch.equal` to one process evaluating the whole pool (BASELINE
configs[3] is the same command without --rehearsal and with 100000 candidates).  No N > 1 RCCL run exists; this is not one."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_eight_rank_bench_rehearsal_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_PORT"] = str(29300 + os.getpid() % 150)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--pool-total", "16384", "--rehearsal",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 only, one line
    r = json.loads(lines[0])
    assert r["rehearsal"] is True and r["value"] is None and r["n_gpus"] == 8 and r["scaling"] == "strong"
    rep = r["rehearsal_report"]
    assert rep["ranks"] == 8 and rep["devices"] == 1 and rep["backend"] == "gloo"
    assert rep["gathered"] == 16384 and rep["gather_equals_single_process"] is True, rep
    assert r["same_results_without_delta_cache"] is True and r["same_results_ranks"] == [True] * 8
    blocks = r["shard_blocks"]
    assert blocks[0][0] == 0 and blocks[-1][1] == 16384 and all(b[0] % 32 == 0 for b in blocks)
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(7))
    assert r["config"]["pairs_per_step"] == 16384 and r["overlap_maxerr_vs_oracle"] < 1e-4 and r["yaw_exact_rate"] == 1.0
    # the census every multi-rank line carries: ranks seen, backend, RCCL version, distinct devices -- here eight
    # ranks on ONE device, which the line says itself
    d = r["distributed"]
    assert d["ranks_seen"] == list(range(8)) and d["world_size"] == 8 and d["backend"] == "gloo"
    assert d["distinct_devices"] == 1 and d["one_rank_per_gpu"] is False and d["rccl_version"]
    assert "8 rank(s) seen" in p.stderr
