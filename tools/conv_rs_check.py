"""Bit-level check of the row-sharing conv loop: prints one hash per (shape, split form) of the 3x3 conv output through the
C ABI (diff the output of two builds, e.g. the product library and a tools/build_variant.sh side library); also checks
against torch."""
import sys, os, math, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
lib = _lib.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, hw, cin, cout in [(30, 64, 320, 320), (30, 64, 640, 320), (33, 32, 640, 640), (57, 16, 1280, 1280), (120, 8, 1280, 1280),
                         (24, 32, 128, 128), (9, 128, 64, 128), (5, 256, 64, 64)]:
    M, K = B * hw * hw, 9 * cin
    x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
    W = (torch.randn(cout, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(cout, device=dev)
    R = torch.randn(M, cout, device=dev).to(torch.bfloat16)
    for splits in (0, -4, 4):
        out = torch.zeros(M, cout, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, cout, K, abs(splits)), 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.hedit_k_gemm(_lib.ptr(x), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(R), _lib.ptr(out), M, cout, K,
                                    cin, cout, cout, 1, hw, hw, cin, hw, hw, splits, _lib.ptr(ws), None))
        torch.cuda.synchronize()
        h = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
        msg = ""
        if splits == 0 and M * cout * K < 3e13:
            w4 = W.float().view(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()
            ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w4, bias, padding=1).permute(0, 2, 3, 1).reshape(M, cout)
            ref = ref.to(torch.bfloat16).float() + R.float()
            err = (out.float() - ref).norm() / ref.norm()
            msg = f" rel err vs torch fp32 {err:.2e}"
        print(f"B={B} {hw}x{hw} {cin}->{cout} splits={splits:2d} {h}{msg}", flush=True)

# mode 3: 3x3 on the 2x nearest-upsampled image (the UNet's upsamplers)
for B, hin, c in [(40, 32, 320), (48, 16, 640), (120, 8, 1280)]:
    ho = 2 * hin
    M, K = B * ho * ho, 9 * c
    x = torch.randn(B, hin, hin, c, device=dev).to(torch.bfloat16)
    W = (torch.randn(c, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(c, device=dev)
    for splits in (0, -4, 4):
        out = torch.zeros(M, c, device=dev, dtype=torch.bfloat16)
        ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, c, K, abs(splits)), 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.hedit_k_gemm(_lib.ptr(x), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, c, K,
                                    c, c, c, 3, hin, hin, c, ho, ho, splits, _lib.ptr(ws), None))
        torch.cuda.synchronize()
        h = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
        msg = ""
        if splits == 0:
            w4 = W.float().view(c, 3, 3, c).permute(0, 3, 1, 2).contiguous()
            up = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
            ref = torch.nn.functional.conv2d(up, w4, bias, padding=1).permute(0, 2, 3, 1).reshape(M, c)
            err = (out.float() - ref).norm() / ref.norm()
            msg = f" rel err vs torch fp32 {err:.2e}"
        print(f"up B={B} {hin}->{ho} {c} splits={splits:2d} {h}{msg}", flush=True)
