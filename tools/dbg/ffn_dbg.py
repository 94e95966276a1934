import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch, torch.nn.functional as F
from hedit import _lib
lib = _lib.lib(); dev = "cuda:0"
C = 320; M = 128
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, C, generator=g) * 1.5 + 0.2).to(torch.bfloat16).to(dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
w1 = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).to(dev)
b1 = (torch.randn(8 * C, generator=g) * 0.5).to(dev)
b2 = torch.zeros(C, device=dev)
xn = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5).to(torch.bfloat16).float()
proj = xn @ w1.to(torch.bfloat16).float().t() + b1
h, gate = proj.chunk(2, dim=-1)
hid = (h * F.gelu(gate))
for blk in range(4):
    w2 = torch.zeros(C, 4 * C, device=dev)
    idx = torch.arange(C, device=dev)
    w2[idx, blk * C + idx] = 1.0
    ws = torch.empty(lib.hedit_k_ffn_stream_bytes(0), dtype=torch.uint8, device=dev)
    bp = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=dev)
    _lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), None, None, _lib.ptr(ws), _lib.ptr(bp), None))
    out = torch.zeros_like(x)
    _lib.check(lib.hedit_k_ffn_fused(_lib.ptr(x), C, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, _lib.ptr(ws), _lib.ptr(bp),
                                     _lib.ptr(b2), _lib.ptr(out), C, M, C, None))
    torch.cuda.synchronize()
    got = out.float() - x.float()
    want = hid[:, blk * C:(blk + 1) * C]
    err = (got - want).abs()
    tol = want.abs() * 2 ** -6 + 0.03
    badm = (err > tol)
    print(f"blk {blk}: rel {((got - want).norm() / want.norm()).item():.3e} bad {int(badm.sum())} / {badm.numel()}")
    if badm.any():
        rows = badm.any(dim=1).nonzero().flatten().tolist()
        cols = badm.any(dim=0).nonzero().flatten().tolist()
        print("  bad rows:", rows[:40], "n", len(rows))
        print("  bad cols (hidden local):", cols[:64], "n", len(cols))
        r0, c0 = badm.nonzero()[0].tolist()
        print("  sample", r0, c0, got[r0, c0].item(), want[r0, c0].item(), "h", h[r0, blk * C + c0].item(), "gate", gate[r0, blk * C + c0].item())
