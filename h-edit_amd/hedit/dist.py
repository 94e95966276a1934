"""Multi-GPU plumbing: one process per GPU, independent images sharded across ranks.

The reference has no distributed code (one process, one GPU, one image at a time,
text-guided/main_p2p.py:87,110).  Images are independent, so the build shards the image list and
needs exactly one collective: a broadcast of the weights from rank 0 at start-up (RCCL over xGMI
on MI355X: backend "nccl"; the same code runs on "gloo" for the CPU tests).  No steady-state
traffic; timings are max-reduced over ranks.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard(n_items, rank, world):
    """Contiguous, balanced partition: the indices of `range(n_items)` owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def init_from_env(device=None):
    """Join the job torch.distributed.run started (RANK / WORLD_SIZE / MASTER_* in the environment): backend "nccl"
    (= RCCL) for a CUDA device, "gloo" otherwise.  -> (rank, world); a single process needs no group."""
    rank, world, _ = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        cuda = device is not None and torch.device(device).type == "cuda"
        kw = {"device_id": torch.device(device)} if cuda else {}
        dist.init_process_group(backend="nccl" if cuda else "gloo", rank=rank, world_size=world, **kw)
    return rank, world


def _numel(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


def _agree(rank, src, group, problem, extra):
    """Rank `src` tells every rank how its preparation went BEFORE anybody enters a tensor collective: a rank that
    raised on its own (missing file, missing key, wrong shape) would leave the others blocked in `dist.broadcast` until
    the backend's timeout, with no message.  -> the list of checkpoint keys nobody asked for; raises on EVERY rank when
    `src` reported a problem."""
    box = [problem, list(extra)] if rank == src else [None, None]
    dist.broadcast_object_list(box, src=src, group=group)
    if box[0] is not None:
        raise RuntimeError(f"rank {src} could not provide the weights: {box[0]}")
    return box[1]


class UnexpectedKey:
    """Stands in, on the receiving ranks, for a checkpoint tensor that rank 0 read and no module asked for: its NAME is
    what a strict `load_state_dict` reports, so several ranks refuse the same checkpoints a single process refuses."""
    shape = ()

    def __repr__(self):
        return "<checkpoint key without a parameter>"


def broadcast_state_dict(shapes, sd, src=0, device="cpu", group=None, bf16_names=(), problem=None):
    """Every rank returns the full {name: tensor} dict that rank `src` holds in `sd` (other ranks pass None).
    The tensors are packed, in the deterministic order of `shapes`, into at most TWO flat blobs -- bf16 for the
    names in `bf16_names` (the matrices / convolutions the executor keeps as unscaled bf16 copies anyway:
    UNet2DConditionModel.bf16_exact; rounding them before the transfer changes no packed bit), fp32 for the rest
    (biases, norm parameters, pre-scaled projections) -- and each blob is broadcast once: 1.7 GB instead of 3.4 GB for
    the SD-1.5 UNet, one collective per blob, no per-tensor launches.  Every tensor starts on a 16-byte
    boundary of its blob.  The returned tensors are VIEWS of the two blobs: consume them (load_state_dict) and drop
    the dict as a whole -- keeping a single view alive pins its entire blob.

    Before the blobs, `src` announces whether it has every tensor in the right shape (`problem`: what already went
    wrong while reading, if anything) and which keys of `sd` are not in `shapes`; a failure raises on every rank, and
    the surplus keys come back as `UnexpectedKey` entries behind the real ones, so a strict load sees them everywhere."""
    rank = dist.get_rank(group)
    bf16_names = set(bf16_names)
    from . import _lib
    st16 = _lib.storage_dtype()       # bfloat16, or half in the HEDIT_STORAGE=f16 build: the executor's own 16-bit copies
    plans = {st16: [], torch.float32: []}
    totals = {st16: 0, torch.float32: 0}
    for name, shape in shapes.items():
        dt = st16 if name in bf16_names else torch.float32
        n = _numel(shape)
        plans[dt].append((name, tuple(shape), totals[dt], n))
        align = 16 // (2 if dt == st16 else 4)
        totals[dt] += (n + align - 1) // align * align
    extra = []
    if rank == src and problem is None:
        if sd is None:
            problem = "no state dict on the source rank"
        else:
            missing = [n for n in shapes if n not in sd]
            bad = [f"{n}: {tuple(sd[n].shape)} != {tuple(shapes[n])}" for n in shapes if n in sd and tuple(sd[n].shape) != tuple(shapes[n])]
            if missing:
                problem = f"missing {missing[:5]} ({len(missing)})"
            elif bad:
                problem = f"shape mismatch {bad[:5]} ({len(bad)})"
            extra = [n for n in sd if n not in shapes]
    extra = _agree(rank, src, group, problem, extra)
    out = {}
    for dt, plan in plans.items():
        if not plan:
            continue
        flat = torch.zeros(totals[dt], dtype=dt, device=device)
        if rank == src:
            for name, shape, off, n in plan:
                flat[off:off + n].copy_(sd[name].reshape(-1).to(dtype=dt))
        dist.broadcast(flat, src=src, group=group)
        for name, shape, off, n in plan:
            out[name] = flat[off:off + n].view(shape)
    res = {name: out[name] for name in shapes}
    for name in extra:
        res[name] = sd[name] if rank == src else UnexpectedKey()
    return res


def state_dict_from_rank0(read_fn, shapes, device="cpu", bf16_names=(), group=None):
    """The product drivers' checkpoint path: ONLY rank 0 calls ``read_fn()`` (reads the file), every rank gets the
    tensors by broadcast -- one file read per job instead of one per GPU (replaces the per-process
    StableDiffusionPipeline.from_pretrained of text-guided/main_p2p.py:119 when the job has several ranks).
    A single process (no group) just reads.  If the read fails on rank 0, every rank raises with its message."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return read_fn()
    sd, problem = None, None
    if dist.get_rank(group) == 0:
        try:
            sd = read_fn()
        except Exception as e:      # noqa: BLE001 -- whatever it was, the other ranks must hear about it
            problem = f"{type(e).__name__}: {e}"
    return broadcast_state_dict(shapes, sd, src=0, device=device, group=group, bf16_names=bf16_names, problem=problem)


def max_over_ranks(value, device="cpu", group=None):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_counts(count, device="cpu", group=None):
    """Sum of per-rank processed-unit counts (whole-job throughput numerator)."""
    t = torch.tensor([float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
