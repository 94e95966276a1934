"""CPU, world_size 2, gloo: the N>1 path of bench.py (image sharding, weight broadcast from
rank 0, max-over-ranks timing, whole-job unit count)."""
import os
import socket

import numpy as np
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


def _driver_worker(rank, world, port, q, ckpt_dir):
    """the product drivers' start-up: only rank 0 may touch the checkpoint, the others receive it (bf16 blob for the
    tensors named bf16-exact, fp32 blob for the rest)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rk, w = HD.init_from_env("cpu")
    try:
        from hedit import checkpoint as CK
        shapes = {"m.weight": (6, 5), "m.bias": (6,), "conv.weight": (3, 2, 3, 3), "q.weight": (4, 4), "n.weight": (7,)}
        reads = []

        def read():
            if rank != 0:
                raise AssertionError("a non-zero rank read the checkpoint")
            reads.append(1)
            return CK.read_component(ckpt_dir)[1]
        got = HD.state_dict_from_rank0(read, shapes, device="cpu", bf16_names={"m.weight", "conv.weight"})
        surplus = [k for k in got if k not in shapes]
        # numpy payloads: a torch tensor on an mp.Queue travels as a file descriptor the parent must fetch from THIS process
        # while it is still alive (rebuild_storage_fd) -- under load the worker had exited first.  ndarrays are pickled by value.
        q.put((rank, {k: (str(got[k].dtype), got[k].float().numpy().copy(), got[k].data_ptr() % 16) for k in shapes}, len(reads),
               (rk, w), list(got), surplus))
    finally:
        dist.destroy_process_group()


def test_driver_reads_on_rank0_and_broadcasts(tmp_path):
    from hedit import checkpoint as CK
    g = torch.Generator().manual_seed(1)
    shapes = {"m.weight": (6, 5), "m.bias": (6,), "conv.weight": (3, 2, 3, 3), "q.weight": (4, 4), "n.weight": (7,)}
    sd = {k: torch.randn(v, generator=g) for k, v in shapes.items()}
    sd["unused.extra"] = torch.zeros(3)
    CK.write_component(str(tmp_path), {"x": 1}, sd)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, a, reads0, rw0, keys0, sur0), (_, b, reads1, rw1, keys1, sur1) = res
    assert reads0 == 1 and reads1 == 0 and rw0 == (0, 2) and rw1 == (1, 2)
    assert list(a) == list(shapes) == list(b)
    # the checkpoint key no module asked for is reported on BOTH ranks, behind the real ones: a strict load_state_dict
    # refuses the same checkpoints on several ranks as in a single process
    assert keys0 == keys1 == list(shapes) + ["unused.extra"] and sur0 == sur1 == ["unused.extra"]
    for k in shapes:
        assert a[k][0] == b[k][0] == ("torch.bfloat16" if k in ("m.weight", "conv.weight") else "torch.float32")
        assert np.array_equal(a[k][1], b[k][1])                    # every rank holds the same bits
        want = sd[k].to(torch.bfloat16).float() if k in ("m.weight", "conv.weight") else sd[k]
        assert np.array_equal(a[k][1], want.numpy())               # bf16 only where it was asked for; fp32 tensors exact
        assert a[k][2] == 0 and b[k][2] == 0                       # 16-byte aligned views


def _failing_worker(rank, world, port, q, what):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    HD.init_from_env("cpu")
    try:
        shapes = {"m.weight": (6, 5), "m.bias": (6,)}

        def read():
            if what == "io":
                raise FileNotFoundError("no such checkpoint: /nowhere/unet")
            sd = {"m.weight": torch.zeros(6, 5), "m.bias": torch.zeros(6)}
            if what == "missing":
                del sd["m.bias"]
            if what == "shape":
                sd["m.weight"] = torch.zeros(5, 6)
            return sd
        try:
            HD.state_dict_from_rank0(read, shapes, device="cpu")
            q.put((rank, None))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_rank0_failure_reaches_every_rank():
    """A checkpoint problem on rank 0 (file, key, shape) raises on BOTH ranks with rank 0's message instead of leaving
    rank 1 blocked in the broadcast until the backend times out."""
    ctx = mp.get_context("spawn")
    for what, needle in (("io", "FileNotFoundError"), ("missing", "missing ['m.bias']"), ("shape", "shape mismatch")):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q, what)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, msg in res:
            assert msg is not None and needle in msg and "rank 0" in msg, (what, rank, msg)


def test_shard_properties():
    for n in (0, 1, 7, 8, 128):
        for world in (1, 2, 3, 8):
            parts = [HD.shard(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
