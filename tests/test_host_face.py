"""The PRODUCT's face-swapping host code (hedit/inversion/sde_inversion.py, h_edit_R.py: pure torch around a callable
eps-network) against the vectors produced by running the reference (g11), with the pinned CPU restatement of the pixel
UNet standing in for the HIP executor -- no GPU needed."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.tiny import TinyIdLoss, TinyLpips, hash_normal  # noqa: E402
from test_oracle_face import CASES, G11, face_state_dict, linear_betas  # noqa: E402
from oracle import ddpm_unet  # noqa: E402
from hedit.inversion.h_edit_R import h_Edit_R  # noqa: E402
from hedit.inversion.sde_inversion import inversion_forward_process_sde  # noqa: E402

torch.set_num_threads(4)


@pytest.fixture(scope="module")
def model():
    m = ddpm_unet.Model(**ddpm_unet.TINY_DDPM).eval()
    m.load_state_dict(face_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def test_product_sde_inversion_matches_reference(model):
    vec = np.load(G11)
    T = 10
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    x0 = hash_normal((1, 3, 32, 32), 654) * 0.6
    _, zs, xts, noise = inversion_forward_process_sde(model, x0, linear_betas(), seq, etas=1.0, num_inference_steps=T, device="cpu")
    assert np.allclose(xts.numpy(), vec["xts"], atol=2e-5)
    assert np.allclose(zs.numpy(), vec["zs"], atol=5e-4, rtol=1e-4)
    assert noise.shape == xts.shape and float(noise[0].abs().max()) == 0.0


@pytest.mark.parametrize("name,skip,K,w,use_id,use_lp,use_mask", CASES, ids=[c[0] for c in CASES])
def test_product_face_loop_matches_reference(model, name, skip, K, w, use_id, use_lp, use_mask):
    vec = np.load(G11)
    T = 10
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    zs, xts = torch.from_numpy(vec["zs"]), torch.from_numpy(vec["xts"])
    after = T - skip
    mask = torch.from_numpy(vec["mask"]) if use_mask else None
    out = h_Edit_R(model, TinyLpips() if use_lp else None, TinyIdLoss() if use_id else None, xts[after].clone(), linear_betas(),
                   seq, eta=1.0, zs=zs[:after], weight_edit_face=w, optimization_steps=K, after_skip_steps=after,
                   num_inference_steps=T, soft_face_mask=mask)
    assert out.shape == (1, 3, 32, 32)
    assert np.allclose(out.detach().numpy(), vec[name], atol=5e-4, rtol=1e-4)
