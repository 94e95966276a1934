"""Self-attention determinism probe: identical batch rows must give bit-identical outputs, run to run."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import ctypes as C
import torch
from hedit import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
for (N, c, heads, B) in [(4096, 320, 8, 4), (1024, 640, 8, 4), (256, 1280, 8, 4), (1024, 64, 2, 6), (256, 128, 2, 4)]:
    d = c // heads
    g = torch.Generator().manual_seed(1)
    q1 = torch.randn(1, N, c, generator=g) * (d ** -0.5) * 1.5 * 2.0
    k1 = torch.randn(1, N, c, generator=g) * 2.0
    v1 = torch.randn(1, N, c, generator=g)
    qk = torch.cat([q1, k1], -1).repeat(B, 1, 1).reshape(B * N, 2 * c).to(torch.bfloat16).to(dev).contiguous()
    vt = v1.repeat(B, 1, 1).reshape(B * N, c).t().contiguous().to(torch.bfloat16).to(dev)
    kv = qk[:, c:]
    outs = []
    for rep in range(3):
        out = torch.zeros(B * N, c, device=dev, dtype=torch.bfloat16)
        _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * c, C.c_void_p(kv.data_ptr()), 2 * c, _lib.ptr(vt), B * N,
                                         _lib.ptr(out), c, B, N, heads, d, None, None, None))
        torch.cuda.synchronize()
        outs.append(out.reshape(B, N, c).clone())
    rows = max((outs[0][b].float() - outs[0][0].float()).abs().max().item() for b in range(1, B))
    runs = max((outs[r].float() - outs[0].float()).abs().max().item() for r in range(1, 3))
    bad = (outs[0][1] != outs[0][0]).nonzero()
    print(f"N={N} d={d} heads={heads}: max |row b - row 0| = {rows:.3e}, max |run r - run 0| = {runs:.3e}, "
          f"mismatching elements row1 vs row0: {bad.shape[0]}", bad[:5].tolist())
