"""-m gpu: BASELINE configs[4] (text + style) at its REAL shapes against the oracle:
the SD-1.x autoencoder at a 64 x 64 latent / 512 x 512 image (oracle/sd_vae.py, fp32 on the host), the ViT-B/16 style
prefix (oracle/reward_nets.py, pinned on the reference's CLIPEncoder through tests/golden/g10) and the composed style
update of text-guided-n-style/inversion/h_edit.py:162-182 (oracle/loops.py::_style_step).  Synthetic weights, identical
on both sides.  The toy-size twins of these tests live in test_gpu_vae.py / test_gpu_style.py; the adjoint identity of
test_gpu_vae.py::test_decode_vjp_sd15_shape_adjoint_identity is a self-consistency check, these are parity.

Tolerances (relative L2, bf16 activations with fp32 accumulation on the GPU vs fp32 on the host), with the values
measured on MI355X in the docstrings."""
import copy
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gpu as G  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd_vae_pair():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hedit.vae import AutoencoderKL
    from oracle import sd_vae
    hip = AutoencoderKL(device=G.dev())
    sd = hip.init_random(1)
    om = sd_vae.AutoencoderKL()
    om.load_state_dict(sd)
    om.eval()
    for p in om.parameters():
        p.requires_grad_(False)
    return hip, om


def test_sd15_decode_and_encode_512_match_oracle(sd_vae_pair):
    """(1,4,64,64) latent -> (1,3,512,512) image and back to the mode of the posterior, both against oracle/sd_vae.py
    (reference call sites text-guided/main_p2p.py:159 encode(...).latent_dist.mode(), :263 decode; n-style h_edit.py:170).
    Measured on MI355X: decode 9.2e-3, encode 1.3e-2."""
    hip, om = sd_vae_pair
    g = torch.Generator().manual_seed(9)
    z = torch.randn(1, 4, 64, 64, generator=g)
    with torch.no_grad():
        want = om.decode(z / 0.18215).sample
    got = hip.decode(z.to(G.dev()) / 0.18215).sample
    G.sync()
    assert got.shape == want.shape == (1, 3, 512, 512) and torch.isfinite(got).all()
    e_dec = G.rel_err(got, want)
    x = want.clamp(-1, 1)
    with torch.no_grad():
        want_lat = om.encode(x).latent_dist.mode()
    got_lat = hip.encode(x.to(G.dev())).latent_dist.mode()
    G.sync()
    assert got_lat.shape == want_lat.shape == (1, 4, 64, 64)
    e_enc = G.rel_err(got_lat, want_lat)
    print(f"SD-1.x VAE at 512^2 vs oracle: decode {e_dec:.3e}, encode mode {e_enc:.3e}")
    G.within(e_dec, 2.5e-2, what='VAE decode 512^2')
    G.within(e_enc, 2.5e-2, what='VAE encode 512^2')


def test_sd15_decode_vjp_512_matches_oracle_autograd(sd_vae_pair):
    """d_z = J^T d_image of the full-size decoder from the HIP backward pass (hedit_vae_decode_vjp) against
    torch.autograd.grad through the fp32 oracle -- how the reference's style closure obtains it (n-style h_edit.py:176-179).
    d_image is the gradient of a smooth functional of the image (its squared distance to a fixed target), like the
    closure's.  Measured on MI355X: 7.4e-3."""
    hip, om = sd_vae_pair
    g = torch.Generator().manual_seed(4)
    z = torch.randn(1, 4, 64, 64, generator=g) / 0.18215
    target = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    zz = z.clone().requires_grad_(True)
    img = om.decode(zz).sample
    d_img = (img.detach() - target)
    (want,) = torch.autograd.grad((img * d_img).sum(), zz)
    got = hip.decode_vjp(z.to(G.dev()), d_img.to(G.dev()))
    G.sync()
    assert got.shape == z.shape and torch.isfinite(got).all()
    err = G.rel_err(got, want)
    print(f"SD-1.x decoder VJP at 512^2 vs oracle autograd: {err:.3e}")
    G.within(err, 4e-2, what='decoder VJP 512^2')


class _OracleStyleEncoder:
    """what oracle/loops.py::_style_step calls, served by the oracle's ViT restatement on the host"""

    def __init__(self, twin):
        self.twin = twin

    def get_gram_matrix_residual(self, img):
        from oracle import reward_nets as RN
        return RN.clip_gram_residual(self.twin, img)


def test_style_step_at_sd_shape_matches_oracle(sd_vae_pair):
    """ONE style update at configs[4]'s shapes on identical eps / latent inputs: Tweedie x0 at t-1 -> 512 x 512 decode ->
    bicubic resize to 224 -> ViT-B/16 prefix -> |Gram residual|_F -> gradient through encoder and decoder -> x - rho g with
    rho = rms(correction) / rms(g) * weight (n-style h_edit.py:162-182), HIP (engine.style_step: hedit_step_tweedie,
    hedit_vae_decode + _vjp, hedit_vit_gram_fwd_bwd, hedit_step_style) against oracle/loops.py::_style_step with the
    oracle autoencoder and style encoder.  The update renormalises g, so what is compared is the step (its direction)
    and the result.  Measured on MI355X: step 1.1e-2, result 1.1e-2."""
    from oracle import loops as OL
    from hedit.clip_guidance import CLIPEncoder
    from hedit.clip_guidance.base_clip import ClipVisualPrefix
    from hedit.engine import HEditEngine
    from hedit.scheduler import DDIMScheduler
    hip_vae, om_vae = sd_vae_pair
    dev = G.dev()
    clip = ClipVisualPrefix().init_random(13)                      # ViT-B/16 shape
    ref = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(17))
    nat = CLIPEncoder(clip_model=clip.float(), device=dev)
    nat.set_reference(ref.to(dev))
    twin = CLIPEncoder(clip_model=copy.deepcopy(nat.clip_model).cpu().float())
    twin.set_reference(ref.clone())

    sch = DDIMScheduler()
    sch.set_timesteps(50)
    hip = types.SimpleNamespace(unet=types.SimpleNamespace(device=dev), vae=hip_vae, scheduler=sch)
    om = types.SimpleNamespace(vae=om_vae, scheduler=sch)

    g = torch.Generator().manual_seed(5)
    e_u, e_cs, e_ct = (torch.randn(1, 4, 64, 64, generator=g) for _ in range(3))
    x = torch.randn(1, 4, 64, 64, generator=g)
    cfg = [1.0, 5.0, 7.5]
    tt = int(sch.timesteps[30])
    e_hat = e_u + cfg[1] * (e_cs - e_u)
    e_tar = e_u + cfg[2] * (e_ct - e_u)
    want = OL._style_step(om, _OracleStyleEncoder(twin), x, e_tar, e_tar - e_hat, tt, 0.55)
    got = HEditEngine(hip).style_step(G.f32(e_u), G.f32(e_cs), G.f32(e_u), G.f32(e_ct), G.f32(x), tt, cfg, nat, 0.55)
    G.sync()
    assert torch.isfinite(got).all()
    e_step, e_res = G.rel_err(got - G.f32(x), want - x), G.rel_err(got, want)
    print(f"SD-shape style step vs oracle: step {e_step:.3e}, result {e_res:.3e}")
    G.within(e_step, 6e-2, what='style step')
    G.within(e_res, 6e-2, what='style result')
