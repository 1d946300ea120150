"""CPU test of the N > 1 path: two processes, gloo backend, the same sharding + summary gather that
bench.py / a Monte-Carlo sweep use over RCCL on GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from corenav_gp_amd import sharding


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(512, 3, 8) == (192, 256)      # BASELINE configs[2]: 64 fits per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(4, 4, 4)


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = sharding.shard_range(n_total, rank, world)
    idx = torch.arange(a, b, dtype=torch.float64)
    local = torch.stack([-100.0 - idx, 0.1 + 0.001 * idx, (idx == 3).to(torch.float64)], 1)   # fit 3 "failed"
    table = sharding.gather_summaries(local, n_total)
    stats = sharding.ensemble_stats(table)
    dist.barrier()
    q.put((rank, table.tolist(), stats))      # plain lists: a tensor in a Queue needs the sender alive until it is read
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8])
def test_gather_summaries_world2(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = torch.arange(n_total, dtype=torch.float64)
    expect = torch.stack([-100.0 - idx, 0.1 + 0.001 * idx, (idx == 3).to(torch.float64)], 1)
    for rank, table, stats in got:
        table = torch.tensor(table, dtype=torch.float64)
        assert torch.equal(table, expect)                      # global fit order on every rank
        assert stats["n"] == n_total and stats["n_failed"] == 1
        ok = expect[:, 2] == 0
        assert stats["logml_mean"] == pytest.approx(float(expect[ok, 0].mean()))
        assert stats["max_sigma"] == pytest.approx(float(expect[ok, 1].max()))
