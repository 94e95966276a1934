"""Identity-reward (IR-SE50, torch) forward + input-gradient time per call at batch 8: fp32 vs bf16 autocast vs channels_last."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.arcface import IDLoss
dev = "cuda:0"
idl = IDLoss(ref=torch.randn(1, 3, 256, 256) * 0.4, device=dev, seed=1)
x = (torch.randn(8, 3, 256, 256, device=dev) * 0.4)
def run(mode):
    xx = x.clone().requires_grad_(True)
    if mode == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            l = idl.get_cosine_loss(xx)
    else:
        l = idl.get_cosine_loss(xx)
    (g,) = torch.autograd.grad(l, xx)
    return g
for mode in ("fp32", "bf16"):
    for _ in range(3): run(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g = run(mode)
    torch.cuda.synchronize(); print(mode, "%.2f ms / call" % ((time.perf_counter() - t0) / 10 * 1e3), float(g.abs().max()))
g32 = run("fp32"); g16 = run("bf16")
print("rel diff bf16 vs fp32 grad:", float((g16.float() - g32).norm() / g32.norm()))
idl.facenet.to(memory_format=torch.channels_last)
for _ in range(3): run("fp32")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run("fp32")
torch.cuda.synchronize(); print("fp32 channels_last %.2f ms / call" % ((time.perf_counter() - t0) / 10 * 1e3))
# --- captured into a HIP graph (static shapes): loss + input gradient replayed without the Python / autograd launch path
idl.facenet.to(memory_format=torch.contiguous_format)
static_x = x.clone().requires_grad_(True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        l = idl.get_cosine_loss(static_x); torch.autograd.grad(l, static_x)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    l = idl.get_cosine_loss(static_x)
    (static_g,) = torch.autograd.grad(l, static_x)
gr.replay(); torch.cuda.synchronize()
print("graph grad equals eager:", float((static_g - g32).norm() / g32.norm()))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): gr.replay()
torch.cuda.synchronize(); print("graph replay %.2f ms / call" % ((time.perf_counter() - t0) / 20 * 1e3))
