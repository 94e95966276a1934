"""Pixel UNet (CelebA-HQ 256 configuration, random weights): ms per eps evaluation with the GroupNorm statistics pass over
every tensor against the statistics taken in the producing convolution's epilogue (csrc/gnstat.h; hedit_test_set_flags
bit 2 flips the build's default).  `python tools/gn_fuse_bench.py [batch ...]`, default batches 8 and 1."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit import _lib  # noqa: E402
from hedit.diffusion import Model  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:]] or [8, 1]
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    hip = Model(device=dev)
    hip.init_random(1)
    for B in batches:
        x = torch.randn(B, 3, 256, 256, device=dev) * 0.8
        res = {}
        for flag in (0, 4, 0, 4):
            _lib.check(lib.hedit_test_set_flags(flag))
            try:
                hip(x, 501.0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    out = hip(x, 501.0)
                e1.record()
                torch.cuda.synchronize()
            finally:
                lib.hedit_test_set_flags(0)
            res.setdefault(flag, []).append(e0.elapsed_time(e1) / 5)
            res[("out", flag)] = out
        rel = ((res[("out", 4)] - res[("out", 0)]).norm() / res[("out", 0)].norm()).item()
        print(f"B={B}: default path {min(res[0]):.2f} ms, flipped path {min(res[4]):.2f} ms per evaluation "
              f"({min(res[0]) / min(res[4]):.3f}x), rel. distance of the two eps {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
