"""Magnitudes along the style closure at SD / ViT-B/16 shape with synthetic weights (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.vae import AutoencoderKL
from hedit.clip_guidance import CLIPEncoder
from hedit.clip_guidance.base_clip import ClipVisualPrefix
dev = "cuda:0"
vae = AutoencoderKL(device=dev); vae.init_random(11)
clip = ClipVisualPrefix().init_random(13).half()
enc = CLIPEncoder(clip_model=clip, device=dev)
enc.set_reference(torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(17)).to(dev))
z = (torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(1)) * 5).to(dev).requires_grad_(True)
img = vae.decode(z).sample
print("img", img.abs().max().item(), img.std().item())
feat = enc.clip_model.block_features(enc.preprocess(torch.nn.functional.interpolate(img[:1], size=(224, 224), mode="bicubic")))
print("feat", feat.dtype, feat.float().abs().max().item(), torch.isfinite(feat).all().item())
res = enc.get_gram_matrix_residual(img[:1])
loss = torch.linalg.norm(res)
(g,) = torch.autograd.grad(loss, z)
print("res", res.dtype, res.abs().max().item(), "loss", loss.item(), "g", g.abs().max().item(), torch.isfinite(g).all().item())
