"""Clip sharding and the single all-gather of class scores: pure partition logic plus a world_size-2
gloo run on the CPU (the N > 1 path of bench.py uses the same functions over RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from video_analytics_amd import dist as vdist


def test_shard_ranges_partition_the_clips():
    assert vdist.shard_size(13320, 8) == 1665
    for n, world in [(13320, 8), (10, 4), (3, 8), (64, 1), (65, 2)]:
        got = []
        for r in range(world):
            lo, hi = vdist.shard_range(n, r, world)
            assert 0 <= lo <= hi <= n and hi - lo <= vdist.shard_size(n, world)
            got += list(range(lo, hi))
        assert got == list(range(n))


def test_gather_is_identity_for_one_process():
    x = torch.arange(12.0).reshape(6, 2)
    assert torch.equal(vdist.gather_scores(x, 6, world=1), x)


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = vdist.init(backend="gloo")
    lo, hi = vdist.shard_range(n_items, r, w)
    local = torch.stack([torch.full((2, 101), float(i)) + torch.arange(101.0) * 1e-3 for i in range(lo, hi)]) \
        if hi > lo else torch.zeros((0, 2, 101))
    out = vdist.gather_scores(local, n_items, w)
    t = vdist.max_over_ranks(float(r + 1), torch.device("cpu"))
    vdist.barrier()
    q.put((r, out[:, 0, 0].tolist(), tuple(out.shape), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_gloo_world2_gather(n_items):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, firsts, shape, t in res:
        assert shape == (n_items, 2, 101)
        assert firsts == [float(i) for i in range(n_items)]  # global clip order, padding dropped
        assert t == 2.0
