"""Helpers for the -m gpu tests: bf16 plumbing and error metrics."""
import ctypes as C

import torch

from hedit import _lib


def dev():
    return torch.device("cuda:0")


def bf(t):
    return t.to(device=dev(), dtype=_lib.storage_dtype()).contiguous()


def f32(t):
    return t.to(device=dev(), dtype=torch.float32).contiguous()


def rel_err(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return ((got - want).norm() / (want.norm() + 1e-30)).item()


def max_err(got, want):
    return (got.double().cpu() - want.double().cpu()).abs().max().item()


def sync():
    torch.cuda.synchronize()


def make_plan(n_pairs=0, pair_src=None, pair_tar=None, singles=None, qk_src=None, mixT=None, bvec=None, mode=1):
    """Build a _lib.P2PPlan from torch tensors; returns (plan, keepalive)."""
    p = _lib.P2PPlan()
    keep = []

    def i32(v):
        t = torch.tensor(v, dtype=torch.int32, device=dev())
        keep.append(t)
        return t.data_ptr()

    p.mode = mode
    p.n_pairs = n_pairs
    p.pair_src = i32(pair_src) if pair_src else None
    p.pair_tar = i32(pair_tar) if pair_tar else None
    p.singles = i32(singles) if singles else None
    p.n_single = len(singles) if singles else 0
    p.qk_src = i32(qk_src) if qk_src is not None else None
    if mixT is not None:
        keep += [mixT, bvec]
        p.mixT = mixT.data_ptr()
        p.bvec = bvec.data_ptr()
    p.h_store = None
    p.n_store = 0
    return p, keep
