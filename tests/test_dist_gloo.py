"""CPU, world_size 2, gloo: the N>1 path of bench.py (image sharding, weight broadcast from
rank 0, max-over-ranks timing, whole-job unit count)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hedit import dist as HD


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = {"a.weight": (4, 3), "b.bias": (5,), "c.weight": (2, 2, 3, 3)}
        sd = None
        if rank == 0:
            g = torch.Generator().manual_seed(0)
            sd = {k: torch.randn(v, generator=g) for k, v in shapes.items()}
        got = HD.broadcast_state_dict(shapes, sd, src=0)
        checksum = float(sum(t.double().sum() for t in got.values()))
        mine = HD.shard(11, rank, world)
        tmax = HD.max_over_ranks(1.0 + rank)
        total = HD.gather_counts(len(mine))
        q.put((rank, checksum, mine, tmax, total))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, m0, t0, n0), (r1, c1, m1, t1, n1) = res
    assert c0 == c1                                   # both ranks hold rank 0's weights
    assert sorted(m0 + m1) == list(range(11)) and not set(m0) & set(m1)
    assert t0 == t1 == 2.0                            # max over ranks
    assert n0 == n1 == 11.0


def test_shard_properties():
    for n in (0, 1, 7, 8, 128):
        for world in (1, 2, 3, 8):
            parts = [HD.shard(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
