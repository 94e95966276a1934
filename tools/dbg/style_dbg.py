import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch, torch.nn.functional as F
from hedit.clip_guidance import CLIPEncoder
from hedit.clip_guidance.base_clip import ClipVisualPrefix
dev = torch.device("cuda:0")
clip = ClipVisualPrefix(width=64, layers=3, heads=1, patch_size=32, input_resolution=224).init_random(3)
enc = CLIPEncoder(clip_model=clip.float(), device=dev)
enc.set_reference(torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(dev))
g = torch.Generator().manual_seed(8)
ims = (torch.randn(3, 3, 64, 64, generator=g) * 0.5).to(dev)
def run(fn):
    x = ims.clone().requires_grad_(True)
    l = fn(x)
    (gr,) = torch.autograd.grad(l.sum(), x)
    return l.detach().clone(), gr.clone()
la, ga = run(enc.gram_residual_norms)
la2, ga2 = run(enc.gram_residual_norms)
lb, gb = run(lambda x: enc.gram_residual_norms_each([enc] * 3, x))
print("repeat: loss eq", torch.equal(la, la2), "grad eq", torch.equal(ga, ga2), ((ga - ga2).norm() / ga.norm()).item())
print("shared vs each: loss eq", torch.equal(la, lb), "grad eq", torch.equal(ga, gb), ((ga - gb).norm() / ga.norm()).item(), la.tolist(), lb.tolist())
# without the torch resize in front: native only
x224 = enc.preprocess(F.interpolate(ims, size=(224, 224), mode="bicubic")).detach()
l1, g1 = enc._native_loss_and_grad(x224)
l2, g2 = enc._native_loss_and_grad(x224, ref=torch.stack([enc._native_ref(dev)] * 3).contiguous())
l3, g3 = enc._native_loss_and_grad(x224)
print("native shared vs per-image ref: loss eq", torch.equal(l1, l2), "grad eq", torch.equal(g1, g2), ((g1 - g2).norm() / g1.norm()).item())
print("native repeat: ", torch.equal(l1, l3), torch.equal(g1, g3))
lo, go = enc._native_loss_and_grad(x224[1:2])
print("native row alone: loss eq", torch.equal(l1[1:2], lo), "grad eq", torch.equal(g1[1:2], go), ((g1[1:2] - go).norm() / go.norm()).item())
