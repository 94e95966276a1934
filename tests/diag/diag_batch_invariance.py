"""GPU diagnostic: which stage of the --batch driver path is not bit-identical to the one-image path?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.engine import HEditEngine  # noqa: E402
from hedit.pipeline import HEditPipeline  # noqa: E402
from hedit.unet import TINY_CONFIG  # noqa: E402
from hedit.vae import TINY_VAE_CONFIG  # noqa: E402

dev = "cuda:0"
vcfg = dict(TINY_VAE_CONFIG)
vcfg.update(block_out_channels=(64, 64, 128, 128))
model = HEditPipeline.from_random(TINY_CONFIG, seed=0, device=dev, text_layers=2, vae_config=vcfg)
g = torch.Generator().manual_seed(0)
x = (torch.rand(3, 3, 256, 256, generator=g) * 2 - 1).to(dev)
enc_all = model.vae.encode(x).latent_dist.mode()
for i in range(3):
    e1 = model.vae.encode(x[i:i + 1]).latent_dist.mode()
    print("encode", i, torch.equal(enc_all[i:i + 1], e1), (enc_all[i:i + 1] - e1).abs().max().item())
z = torch.randn(3, 4, 32, 32, generator=g).to(dev)
dec_all = model.vae.decode(z).sample
for i in range(3):
    d1 = model.vae.decode(z[i:i + 1]).sample
    print("decode", i, torch.equal(dec_all[i:i + 1], d1), (dec_all[i:i + 1] - d1).abs().max().item())
model.scheduler.set_timesteps(4)
eng = HEditEngine(model)
w0 = z * 0.5
prompts = ["a cat sitting on a bench", "a red car", "a tree"]
lat, zs, lats = eng.ddim_inversion(w0, prompts, 1.0)
for i in range(3):
    l1, z1, _ = eng.ddim_inversion(w0[i:i + 1], prompts[i:i + 1], 1.0)
    print("ddim_inv", i, torch.equal(lat[i:i + 1], l1), torch.equal(zs[:, i:i + 1], z1))
