"""numpy restatement of the GroupNorm pair-statistics tree of h-edit_amd/csrc/gnstat.h (from its header comment, not from its
code): for every unit of 128 rows and every pair of adjacent channels of a bf16 tensor,

    piece(row) = (a + b, fma(b, b, a * a))
    T[r]       = ((piece(r) + piece(r + 32)) + piece(r + 64)) + piece(r + 96)        r = 0 .. 31
    unit       = (..((T[0] + T[16]) + (T[1] + T[17])) + ..) + (T[15] + T[31])

all in fp32.  Squares of bf16 values are exact in fp32 (16 significant bits), so fma(b, b, a * a) is the correctly rounded sum
of two exact fp32 numbers = one fp32 addition: the restatement is bit-exact without an fma."""
import numpy as np


def bf16_bits_to_f32(bits, half=False):
    """storage bits -> fp32: bfloat16, or IEEE half for the HEDIT_STORAGE=f16 build (squares of 11-bit mantissas are exact in
    fp32 too, so the restatement below stays bit-exact)"""
    bits = np.asarray(bits, dtype=np.uint16)
    if half:
        return bits.view(np.float16).astype(np.float32)
    return (bits.astype(np.uint32) << 16).view(np.float32)


def pair_stats(bits, half=False):
    """bits: uint16 [M][N] (M % 128 == 0, N % 2 == 0) -> float32 [M // 128][N // 2][2]"""
    M, N = bits.shape
    assert M % 128 == 0 and N % 2 == 0
    v = bf16_bits_to_f32(bits, half).reshape(M // 128, 4, 32, N // 2, 2)  # [unit][i][r][pair][a, b]: row = r + 32 i
    a, b = v[..., 0], v[..., 1]
    s = a + b
    q = b * b + a * a                                                    # (both products exact, one rounding)
    piece = np.stack([s, q], axis=-1).astype(np.float32)                 # [unit][i][r][pair][2]
    T = ((piece[:, 0] + piece[:, 1]) + piece[:, 2]) + piece[:, 3]        # [unit][r 32][pair][2]
    acc = T[:, 0] + T[:, 16]
    for r in range(1, 16):
        acc = acc + (T[:, r] + T[:, r + 16])
    return acc.astype(np.float32)
