"""-m gpu: the native style encoder (hedit_vit_gram / hedit_vit_gram_fwd_bwd, csrc/vit.hip -- SURVEY.md section 8 row
a19) against (1) vectors produced by RUNNING the reference's CLIPEncoder.get_gram_matrix_residual and CLIP ViT
(text-guided-n-style/clip_guidance/base_clip.py, clip/model.py) at toy width, tests/golden/g10_clip.npz, and (2) the
torch fp32 mirror at ViT-B/16 shape.  Tolerances: fp32 token stream with 16 mantissa bits per GEMM operand and fp32
accumulation -> Gram residual / loss to 1e-4 relative, image gradient to 2e-3 relative L2 (the reference itself runs
this encoder in fp16, model.py:414-435)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gpu as G  # noqa: E402
from helpers.tiny import hash_normal  # noqa: E402
from test_host_clip import G10, toy_prefix  # noqa: E402
from hedit.clip_guidance import CLIPEncoder  # noqa: E402
from hedit.clip_guidance.base_clip import ClipVisualPrefix  # noqa: E402


@pytest.fixture(scope="module")
def enc(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    g = np.load(G10)
    path = str(tmp_path_factory.mktemp("style") / "ref.png")
    Image.fromarray(g["ref_rgb"]).save(path)
    return CLIPEncoder(need_ref=True, ref_path=path, clip_model=toy_prefix(), device=G.dev(), backend="hip"), g


@pytest.mark.parametrize("i,hw", [(0, (64, 64)), (1, (96, 80))])
def test_native_gram_residual_and_gradient_match_reference_vectors(enc, i, hw):
    e, g = enc
    im = G.f32(hash_normal((1, 3) + hw, 900 + i) * 0.6).requires_grad_(True)
    res = e.get_gram_matrix_residual(im.detach())
    G.sync()
    assert res.shape == (64, 64)
    assert G.rel_err(res, torch.from_numpy(g[f"residual{i}"])) < 1e-4
    loss = e.gram_residual_norms(im).sum()
    (grad,) = torch.autograd.grad(loss, im)
    G.sync()
    assert abs(loss.item() - g[f"loss{i}"][0]) < 1e-4 * g[f"loss{i}"][0]
    assert G.rel_err(grad, torch.from_numpy(g[f"grad{i}"])) < 2e-3


def test_native_matches_torch_mirror_at_vit_b16_shape_and_is_batch_invariant():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    m = ClipVisualPrefix().init_random(13)
    ref = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(17))
    nat = CLIPEncoder(clip_model=m.float(), device=G.dev(), backend="hip")
    tor = CLIPEncoder(clip_model=m.float(), device=G.dev(), backend="torch")
    nat.set_reference(ref.to(G.dev()))
    tor.set_reference(ref.to(G.dev()))
    ims = G.f32(torch.randn(4, 3, 512, 512, generator=torch.Generator().manual_seed(3)) * 0.5)
    xs = [ims.clone().requires_grad_(True) for _ in range(2)]
    ln, lt = nat.gram_residual_norms(xs[0]), tor.gram_residual_norms(xs[1])
    gn, gt = torch.autograd.grad(ln.sum(), xs[0])[0], torch.autograd.grad(lt.sum(), xs[1])[0]
    G.sync()
    assert G.rel_err(ln, lt) < 1e-4
    assert G.rel_err(gn, gt) < 2e-3
    one = nat.gram_residual_norms(ims[2:3])
    G.sync()
    assert torch.equal(one, ln[2:3].detach())


def test_backend_hip_has_no_cpu_fallback():
    e = CLIPEncoder(clip_model=toy_prefix(), backend="hip")
    e.set_reference(torch.zeros(1, 3, 224, 224))
    with pytest.raises(RuntimeError):
        e.gram_residual_norms(torch.zeros(1, 3, 64, 64))
