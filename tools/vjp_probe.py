"""Adjoint check of the SD-shaped decoder VJP at several finite-difference steps (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.vae import AutoencoderKL
dev = "cuda:0"
hip = AutoencoderKL(device=dev)
hip.init_random(5)
g = torch.Generator().manual_seed(1)
z = torch.randn(1, 4, 64, 64, generator=g).to(dev)
v = torch.randn(1, 4, 64, 64, generator=g).to(dev)
def jv(eps):
    return (hip.decode(z + eps * v).sample - hip.decode(z - eps * v).sample) / (2 * eps)
u = jv(0.2)
jtu = hip.decode_vjp(z, u)
rhs = (v.double() * jtu.double()).sum().item()
for e in (0.02, 0.05, 0.1, 0.2, 0.3, 0.5):
    j = jv(e)
    print(e, "lhs", (j.double() * u.double()).sum().item(), "|jv|", j.norm().item(), "rhs", rhs)
f0 = hip.decode(z).sample
print("|f|", f0.norm().item(), "f std", f0.std().item())
