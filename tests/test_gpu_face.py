"""-m gpu: the face-swapping path -- the pixel DDPM UNet on the HIP
executor (csrc/ddpm.hip) and the h-Edit-R face loop / SDE inversion on top of it -- against the oracle
that is pinned on the reference's own code (tests/test_oracle_face.py, g11)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gpu as G  # noqa: E402
from helpers.tiny import TinyIdLoss, TinyLpips, hash_normal  # noqa: E402

pytestmark = pytest.mark.gpu


def make_pair(config, seed=0, out_scale=1.0):
    from hedit.diffusion import Model
    from oracle import ddpm_unet
    hip = Model(config, device=G.dev())
    sd = hip.init_random(seed)
    if out_scale != 1.0:
        sd["conv_out.weight"] = sd["conv_out.weight"] * out_scale
        sd["conv_out.bias"] = sd["conv_out.bias"] * out_scale
        hip.load_state_dict(sd)
    keys = ("in_channels", "out_ch", "ch", "ch_mult", "num_res_blocks", "attn_resolutions", "image_size")
    om = ddpm_unet.Model(**{k: hip.config[k] for k in keys}).eval()
    om.load_state_dict(sd)
    for p in om.parameters():
        p.requires_grad_(False)
    return hip, om


@pytest.fixture(scope="module")
def tiny():
    from hedit.diffusion import TINY_DDPM_CONFIG
    return make_pair(TINY_DDPM_CONFIG)


def test_param_inventory_matches_the_reference_class_names():
    from hedit.diffusion import Model
    from oracle import ddpm_unet
    hip = Model(device=G.dev())                       # CelebA-HQ 256 configuration
    want = {k: tuple(v.shape) for k, v in ddpm_unet.Model(**ddpm_unet.CELEBA_HQ).state_dict().items()}
    assert hip.param_shapes == want
    assert sum(int(np.prod(s)) for s in hip.param_shapes.values()) == 113673219


@pytest.mark.parametrize("B,t", [(1, 991.0), (2, 501.0), (3, 1.0)])
def test_unet_matches_oracle(tiny, B, t):
    hip, om = tiny
    x = hash_normal((B, 3, 32, 32), 10 + B) * 0.8
    with torch.no_grad():
        want = om(x, torch.ones(B) * t)
    got = hip(x.to(G.dev()), (torch.ones(B) * t).to(G.dev()))
    G.sync()
    assert got.shape == want.shape
    G.within(G.rel_err(got, want), 2.5e-2)          # bf16 activations, fp32 accumulation


def test_unet_two_more_levels_and_blocks():
    """a deeper configuration: 3 levels, 2 blocks per level, attention at two resolutions (skip-stack order,
    downsample / upsample at both boundaries, attention inside the down and up paths)"""
    cfg = dict(in_channels=3, out_ch=3, ch=64, ch_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(16, 8), image_size=32)
    hip, om = make_pair(cfg, seed=3)
    x = hash_normal((2, 3, 32, 32), 99) * 0.7
    with torch.no_grad():
        want = om(x, torch.ones(2) * 301)
    got = hip(x.to(G.dev()), 301.0)
    G.sync()
    G.within(G.rel_err(got, want), 2.5e-2)


def test_unet_rejects_mixed_timesteps(tiny):
    hip, _ = tiny
    with pytest.raises(NotImplementedError):
        hip(torch.zeros(2, 3, 32, 32, device=G.dev()), torch.tensor([1.0, 2.0]))


def test_celeba_shape_matches_oracle_and_is_batch_invariant():
    """CelebA-HQ 256 configuration (113.7 M parameters, random weights): eps against the oracle that is pinned on the
    reference's own Model (face-swapping/diffusion/diffusion.py:301-341), repeatable, and a row's eps is a function of
    the row alone -- bit for bit, whatever else shares the batch."""
    from hedit.diffusion import Model
    from oracle import ddpm_unet
    hip = Model(device=G.dev())
    sd = hip.init_random(1)
    x = hash_normal((3, 3, 256, 256), 5) * 0.8
    xg = x.to(G.dev())
    a = hip(xg, 501.0)
    b = hip(xg, 501.0)
    G.sync()
    assert a.shape == (3, 3, 256, 256) and torch.isfinite(a).all()
    assert torch.equal(a, b)
    assert torch.equal(hip(xg[:1], 501.0), a[:1])
    assert torch.equal(hip(xg[1:], 501.0), a[1:])
    om = ddpm_unet.Model(**ddpm_unet.CELEBA_HQ).eval()
    om.load_state_dict(sd)
    with torch.no_grad():
        want = om(x[:1], torch.ones(1) * 501.0)
    G.within(G.rel_err(a[:1], want), 2.5e-2)          # bf16 activations, fp32 accumulation


def linear_betas():
    return torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float64)).float()


def test_sde_inversion_and_face_loop_match_oracle():
    from hedit.diffusion import TINY_DDPM_CONFIG
    from hedit.inversion.h_edit_R import h_Edit_R
    from hedit.inversion.sde_inversion import inversion_forward_process_sde
    from oracle import face_loops
    hip, om = make_pair(TINY_DDPM_CONFIG, seed=2, out_scale=0.3)
    dev = G.dev()
    T = 8
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    betas = linear_betas()
    x0 = hash_normal((1, 3, 32, 32), 654) * 0.6
    zs_o, xts_o = face_loops.sde_inversion(om, x0, betas, seq, etas=1.0, T=T)
    # same chain on the GPU: torch.manual_seed(42) inside both, but CPU and GPU generators differ -> feed the
    # oracle's sampled chain by monkeypatching the sampler is not needed: compare through the defining property
    _, zs_h, xts_h, _ = inversion_forward_process_sde(hip, x0.to(dev), betas.to(dev), seq, etas=1.0, num_inference_steps=T,
                                                      device=dev)
    G.sync()
    assert zs_h.shape == zs_o.shape and torch.isfinite(zs_h).all()
    # edit with the ORACLE's inversion outputs on both sides (identical inputs)
    idl, lp = TinyIdLoss(), TinyLpips()
    import copy
    idl_g, lp_g = copy.deepcopy(idl).to(dev), copy.deepcopy(lp).to(dev)
    for skip, K, w in ((4, 1, 4.0), (4, 2, 4.0), (0, 1, 4.0)):
        after = T - skip
        want = face_loops.h_edit_r_face(om, lp, idl, xts_o[after].clone(), betas, seq, eta=1.0, zs=zs_o[:after],
                                        weight_edit_face=w, optimization_steps=K, after_skip_steps=after,
                                        num_inference_steps=T)
        got = h_Edit_R(hip, lp_g, idl_g, xts_o[after].clone().to(dev), betas.to(dev), seq, eta=1.0, zs=zs_o[:after].to(dev),
                       weight_edit_face=w, optimization_steps=K, after_skip_steps=after, num_inference_steps=T)
        G.sync()
        assert got.shape == (1, 3, 32, 32) and torch.isfinite(got).all()
        # measured 3.6e-3 (4 steps, K = 1 / 2) and 3.1e-2 (8 steps): limits = 2x
        G.within(G.rel_err(got, want), (8e-3 if after <= 4 else 6.5e-2))


def test_unet_batch_grouping_of_the_attention_is_transparent(tiny):
    """the block-diagonal attention stacks up to 4096 / T images per pass: a batch of 20 at T = 256 runs as groups of
    16 + 4; every image must come out exactly as in a batch of its own"""
    hip, _ = tiny
    x = (hash_normal((20, 3, 32, 32), 71) * 0.8).to(G.dev())
    big = hip(x, 401.0)
    parts = torch.cat([hip(x[i:i + 5], 401.0) for i in range(0, 20, 5)])
    G.sync()
    assert torch.isfinite(big).all()
    assert torch.equal(big, parts)


def test_lockstep_faces_equal_single_runs():
    """h_Edit_R(per_image=True) on two faces at once (xT (2,3,S,S), zs (T,2,3,S,S)) == two single-face runs: the batch-mean
    losses are rescaled so that every face receives its own full gradient.  Bit for bit: the eps-network is
    batch-invariant (asserted first) and so is everything between its evaluations."""
    from hedit.diffusion import TINY_DDPM_CONFIG
    from hedit.inversion.h_edit_R import h_Edit_R
    hip, _ = make_pair(TINY_DDPM_CONFIG, seed=2, out_scale=0.3)
    dev = G.dev()
    T = 6
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1][:T]
    betas = linear_betas().to(dev)
    xT = (hash_normal((2, 3, 32, 32), 5) * 0.9).to(dev)
    zs = (hash_normal((T, 2, 3, 32, 32), 6)).to(dev)
    # The stand-in rewards are torch modules (MIOpen convolutions under autograd): torch picks its kernels by batch size,
    # so a batch-2 evaluation is not bit-identical to two batch-1 evaluations of ITS OWN accord.  What is under test is
    # this library's loop, so the stand-ins evaluate image by image (batch-mean of per-image losses: the same value).
    class PerImage(torch.nn.Module):
        def __init__(self, inner, method):
            super().__init__()
            self.inner = inner
            setattr(self, method, lambda x: torch.stack([getattr(inner, method)(x[j:j + 1]) for j in range(x.shape[0])]).mean())
    idl, lp = PerImage(TinyIdLoss().to(dev), "get_cosine_loss"), PerImage(TinyLpips().to(dev), "get_lpips_loss")
    kw = dict(eta=1.0, weight_edit_face=4.0, optimization_steps=2, after_skip_steps=T, num_inference_steps=T)
    e2 = hip(xT, 501.0)
    assert torch.equal(e2[:1], hip(xT[:1], 501.0)) and torch.equal(e2[1:], hip(xT[1:], 501.0))
    both = h_Edit_R(hip, lp, idl, xT, betas, seq, zs=zs, per_image=True, **kw)
    for i in range(2):
        one = h_Edit_R(hip, lp, idl, xT[i:i + 1], betas, seq, zs=zs[:, i:i + 1], **kw)
        G.sync()
        assert torch.equal(both[i:i + 1], one), i
    assert G.rel_err(both[0], both[1]) > 1e-1
