"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (scene partition, gather of controls in
global scene order, max-over-ranks timing)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _shard_torch as shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.scene_range(rank, world, total)
    counts = [shard.scene_range(r, world, total)[1] - shard.scene_range(r, world, total)[0] for r in range(world)]
    # a "control" that encodes the global scene id, as if computed by this rank's GPU
    u = torch.stack([torch.arange(lo, hi, dtype=torch.float64) * 10 + c for c in range(4)], dim=1)
    allu = shard.gather_controls(u, counts)
    if len(set(counts)) == 1:   # bench.py's path: equal shards, one collective into a preallocated tensor
        out = torch.empty((world * (hi - lo), 4), dtype=torch.float64)
        assert shard.gather_controls(u, out=out) is out and torch.equal(out, allu)
    tmax = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
    dist.barrier()
    q.put((rank, lo, hi, allu.numpy().copy(), tmax))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7, 2048])
def test_two_ranks(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == total       # disjoint cover
    expect = torch.stack([torch.arange(total, dtype=torch.float64) * 10 + c for c in range(4)], dim=1).numpy()
    for r in res:
        assert (r[3] == expect).all()                                             # global scene order
        assert r[4] == 2.0                                                        # slowest rank's clock


def test_single_process_is_passthrough():
    u = torch.ones(3, 4, dtype=torch.float64)
    assert shard.gather_controls(u) is u
    assert shard.max_over_ranks(0.5, torch.device("cpu")) == 0.5
    assert [shard.scene_range(r, 8, 2048) for r in (0, 7)] == [(0, 256), (1792, 2048)]
