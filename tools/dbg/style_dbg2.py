import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import gpu as G
from helpers.models import make_pair
from hedit.unet import TINY_CONFIG
from hedit.vae import AutoencoderKL, TINY_VAE_CONFIG
from hedit.clip_guidance import CLIPEncoder
from hedit.clip_guidance.base_clip import ClipVisualPrefix
from hedit.engine import HEditEngine
hip, om, _ = make_pair(TINY_CONFIG, 8, out_scale=0.3)
hip.vae = AutoencoderKL(TINY_VAE_CONFIG, device=G.dev()); hip.vae.init_random(17)
dev = G.dev()
clip = ClipVisualPrefix(width=64, layers=3, heads=1, patch_size=32, input_resolution=224).init_random(3)
enc = CLIPEncoder(clip_model=clip.float(), device=dev)
enc.set_reference(torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev))
g = torch.Generator().manual_seed(8); n = 3
e_u, e_cs, e_ct, x = (G.f32(torch.randn(n, 4, 32, 32, generator=g)) for _ in range(4))
cfg = [1.0, 5.0, 7.5]; tt = int(hip.scheduler.timesteps[4]); eng = HEditEngine(hip)
def rel(a, b): return ((a - b).norm() / (b - x).norm()).item()
a = eng.style_step(e_u, e_cs, e_u, e_ct, x, tt, cfg, enc, 0.5)
a2 = eng.style_step(e_u, e_cs, e_u, e_ct, x, tt, cfg, enc, 0.5)
b = eng.style_step(e_u, e_cs, e_u, e_ct, x, tt, cfg, [enc] * n, 0.5)
print("a vs a2", rel(a, a2), "a vs b", rel(a, b))
# the pieces: decode + vjp determinism and batch invariance
z = torch.randn(3, 4, 32, 32, generator=g).to(dev)
u = torch.randn(3, 3, 64, 64, generator=g).to(dev)
v1 = hip.vae.decode_vjp(z, u); v2 = hip.vae.decode_vjp(z, u); v3 = hip.vae.decode_vjp(z[1:2], u[1:2])
print("vjp repeat eq", torch.equal(v1, v2), "row alone eq", torch.equal(v1[1:2], v3), ((v1[1:2]-v3).norm()/v3.norm()).item())
zc = z.clone().requires_grad_(True)
with torch.enable_grad():
    img = hip.vae.decode(zc).sample
    l = enc.gram_residual_norms(img).sum()
g1 = torch.autograd.grad(l, zc)[0]
zc = z.clone().requires_grad_(True)
with torch.enable_grad():
    img = hip.vae.decode(zc).sample
    l = enc.gram_residual_norms(img).sum()
g2 = torch.autograd.grad(l, zc)[0]
print("closure grad repeat rel", ((g1 - g2).norm() / g1.norm()).item())
