"""The oracle's pixel DDPM UNet (oracle/ddpm_unet.py) and face-swapping sampler (oracle/face_loops.py) against
vectors produced by RUNNING the reference's face-swapping/diffusion/diffusion.py::Model,
inversion/sde_inversion.py and inversion/h_edit_R.py at toy size (tests/golden/make_golden.py::gen_face, g11)."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.tiny import TinyIdLoss, TinyLpips, hash_normal  # noqa: E402
from oracle import ddpm_unet, face_loops  # noqa: E402

torch.set_num_threads(4)     # as the generator: identical association of the CPU reductions
G11 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_face.npz")


def face_state_dict(shapes):
    sd = {}
    for name, shape in shapes.items():
        v = hash_normal(tuple(shape), zlib.crc32(name.encode()) % 100003)
        if "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * v
        elif len(shape) == 1:
            v = 0.05 * v
        else:
            v = v * float(np.prod(shape[1:])) ** -0.5
        sd[name] = v
    return sd


def linear_betas():
    return torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float64)).float()


@pytest.fixture(scope="module")
def model():
    m = ddpm_unet.Model(**ddpm_unet.TINY_DDPM).eval()
    m.load_state_dict(face_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    for p in m.parameters():
        p.requires_grad_(False)
    return m


@pytest.fixture(scope="module")
def vec():
    return np.load(G11)


def test_state_dict_names_of_the_celeba_model():
    """the restatement exposes the reference class's parameter names at the CelebA-HQ configuration"""
    m = ddpm_unet.Model(**ddpm_unet.CELEBA_HQ)
    sd = m.state_dict()
    for k in ("temb.dense.0.weight", "down.4.attn.1.q.weight", "down.0.downsample.conv.weight", "mid.attn_1.proj_out.bias",
              "up.5.block.2.nin_shortcut.weight", "up.1.upsample.conv.weight", "norm_out.weight", "conv_out.bias"):
        assert k in sd, k
    assert sd["up.5.block.0.conv1.weight"].shape == (512, 1024, 3, 3)
    assert sum(v.numel() for v in sd.values()) == 113673219


@pytest.mark.parametrize("t", [1, 501, 991])
def test_unet_matches_reference(model, vec, t):
    x = hash_normal((2, 3, 32, 32), 321) * 0.8
    with torch.no_grad():
        got = model(x, torch.ones(2) * t)
    assert np.allclose(got.numpy(), vec[f"unet_t{t}"], atol=2e-5, rtol=1e-5)


def test_sde_inversion_matches_reference(model, vec):
    T = 10
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    x0 = hash_normal((1, 3, 32, 32), 654) * 0.6
    zs, xts = face_loops.sde_inversion(model, x0, linear_betas(), seq, etas=1.0, T=T)
    assert np.allclose(xts.numpy(), vec["xts"], atol=2e-5)
    assert np.allclose(zs.numpy(), vec["zs"], atol=5e-4, rtol=1e-4)


CASES = [("face_k2", 0, 2, 4.0, True, True, False), ("face_k1_skip3_mask", 3, 1, 6.0, True, True, True),
         ("face_idonly", 0, 1, 4.0, True, False, False), ("face_lponly", 2, 2, 4.0, False, True, False)]


@pytest.mark.parametrize("name,skip,K,w,use_id,use_lp,use_mask", CASES, ids=[c[0] for c in CASES])
def test_face_loop_matches_reference(model, vec, name, skip, K, w, use_id, use_lp, use_mask):
    T = 10
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    zs, xts = torch.from_numpy(vec["zs"]), torch.from_numpy(vec["xts"])
    after = T - skip
    mask = torch.from_numpy(vec["mask"]) if use_mask else None
    out = face_loops.h_edit_r_face(model, TinyLpips() if use_lp else None, TinyIdLoss() if use_id else None,
                                   xts[after].clone(), linear_betas(), seq, eta=1.0, zs=zs[:after], weight_edit_face=w,
                                   optimization_steps=K, after_skip_steps=after, num_inference_steps=T,
                                   soft_face_mask=mask)
    assert out.shape == (1, 3, 32, 32)
    # fp32 reassociation noise through the normalised-feature gradients: measured <= 2.3e-4 absolute
    assert np.allclose(out.detach().numpy(), vec[name], atol=5e-4, rtol=1e-4)
