"""hedit -- MI355X-native h-Edit sampling path (host side).

Python mirrors of the reference's editing API (text-guided/inversion/p2p_h_edit.py,
text-guided/p2p/*) on top of libhedit_hip.so (HIP kernels for gfx950, C ABI in include/hedit.h).
PyTorch-ROCm is used for device memory, streams and torch.distributed only.  There is no CPU or
eager fallback: importing :mod:`hedit._lib` fails loudly when the HIP library is missing.
"""
__version__ = "0.1.0"
