"""GroupNorm / LayerNorm bandwidth at the UNet's level-0 / level-1 sizes (B rows)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
v = os.environ.get("HEDIT_LIB_VARIANT")
if v:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{v}.so.bin")
lib = _lib.lib(); dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 80

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (hw, c) in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1280), (256, 1280), (256, 2560)]:
    x = torch.randn(B * hw, c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    ws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(B, hw, c), dtype=torch.uint8, device=dev)
    us = t(lambda: _lib.check(lib.hedit_k_groupnorm(_lib.ptr(x), _lib.ptr(y), _lib.ptr(g), _lib.ptr(b), B, hw, c, 32, 1e-5, 1, _lib.ptr(ws), None)))
    byt = x.numel() * 2
    print(f"groupnorm+silu B={B} HW={hw} C={c}: {us:8.1f} us  {3 * byt / us / 1e6:6.2f} TB/s (2 reads + 1 write)")
for (hw, c) in [(4096, 320), (1024, 640), (256, 1280)]:
    x = torch.randn(B * hw, c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    us = t(lambda: _lib.check(lib.hedit_k_layernorm(_lib.ptr(x), _lib.ptr(y), _lib.ptr(g), _lib.ptr(b), B * hw, c, 1e-5, None)))
    byt = x.numel() * 2
    print(f"layernorm      rows={B * hw} C={c}: {us:8.1f} us  {2 * byt / us / 1e6:6.2f} TB/s (1 read + 1 write)")
