"""CPU, world_size 2, gloo: candidate sharding and the single (overlap, yaw) gather of the 1-vs-N sweep."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from overlapnet_amd import distributed as D


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 1024, 100000, 100003):
        for world in (1, 2, 3, 8):
            b = [D.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == D.shard_sizes(n, world)
    with pytest.raises(ValueError):
        D.shard_bounds(10, 2, 2)


def test_pack_unpack_exact():
    ov = torch.tensor([0.0, 1.0, 0.123456789, 1e-30, 0.9999999], dtype=torch.float32)
    yw = torch.tensor([-179, 180, 0, 17, -1], dtype=torch.int32)
    o2, y2 = D.unpack_scores(D.pack_scores(ov, yw))
    assert torch.equal(o2, ov) and torch.equal(y2, yw)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        all_ov = torch.from_numpy(rng.random(n_total).astype(np.float32))
        all_yaw = torch.from_numpy(rng.integers(-179, 181, n_total).astype(np.int32))

        def score(lo, hi):  # stands in for engine.heads on this rank's block of resident candidates
            return all_ov[lo:hi].clone(), all_yaw[lo:hi].clone()

        res = D.sweep_one_vs_n(score, n_total)
        if rank == 0:
            ok = torch.equal(res[0], all_ov) and torch.equal(res[1], all_yaw)
            bm = D.best_match(res[0], res[1], 0.3)
            q.put((ok, bm is not None and bm[0] == int(torch.argmax(all_ov))))
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [0, 1, 5, 1024])
def test_gather_world2_gloo(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, bm_ok = q.get(timeout=10)
    assert ok
    if n_total > 0:
        assert bm_ok
