"""GPU parity of the native AutoencoderKL executor (csrc/vae.hip) against the CPU restatement
(oracle/sd_vae.py) on identical synthetic weights: the steps either side of the editing loop
(reference text-guided/main_p2p.py:159 encode, :263 decode)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gpu as G  # noqa: E402

pytestmark = pytest.mark.gpu


def make_pair(config, seed=0):
    from hedit.vae import AutoencoderKL
    from oracle import sd_vae
    hip = AutoencoderKL(config, device=G.dev())
    sd = hip.init_random(seed)
    om = sd_vae.AutoencoderKL(config)
    om.load_state_dict(sd)
    return hip, om.eval()


@pytest.fixture(scope="module")
def tiny():
    from hedit.vae import TINY_VAE_CONFIG
    return make_pair(TINY_VAE_CONFIG)


def test_param_inventory_matches_oracle_sd15():
    """names and shapes of every parameter == the diffusers-keyed state_dict of the restatement
    (248 tensors, 83,653,863 parameters for the SD-1.x autoencoder)."""
    from hedit.vae import AutoencoderKL
    from oracle import sd_vae
    hip = AutoencoderKL(device=G.dev())
    want = {k: tuple(v.shape) for k, v in sd_vae.AutoencoderKL().state_dict().items()}
    assert hip.param_shapes == want
    assert sum(torch.Size(s).numel() for s in hip.param_shapes.values()) == 83653863


@pytest.mark.parametrize("B,h,w", [(1, 16, 16), (3, 16, 8), (2, 8, 8)])
def test_decode_matches_oracle(tiny, B, h, w):
    hip, om = tiny
    g = torch.Generator().manual_seed(B * 100 + h)
    z = torch.randn(B, 4, h, w, generator=g)
    with torch.no_grad():
        want = om.decode(z).sample
    got = hip.decode(z.to(G.dev())).sample
    G.sync()
    assert got.shape == want.shape == (B, 3, 2 * h, 2 * w)
    G.within(G.rel_err(got, want), 2.5e-2)        # bf16 activations, fp32 accumulation


@pytest.mark.parametrize("B,H,W", [(1, 32, 32), (2, 16, 64)])
def test_encode_mode_matches_oracle(tiny, B, H, W):
    hip, om = tiny
    g = torch.Generator().manual_seed(B * 7 + H)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    with torch.no_grad():
        want = om.encode(x).latent_dist.mode()
    got = hip.encode(x.to(G.dev())).latent_dist.mode()
    G.sync()
    assert got.shape == want.shape == (B, 4, H // 2, W // 2)
    G.within(G.rel_err(got, want), 2.5e-2)


def test_batch_rows_are_independent_and_deterministic(tiny):
    hip, _ = tiny
    g = torch.Generator().manual_seed(3)
    z = torch.randn(1, 4, 16, 16, generator=g).repeat(3, 1, 1, 1).to(G.dev())
    a = hip.decode(z).sample.clone()
    b = hip.decode(z).sample
    G.sync()
    assert torch.equal(a, b)
    assert torch.equal(a[0], a[1]) and torch.equal(a[0], a[2])
    # different rows: each decoded image (and its encoding) is a function of its own latent alone, bit for bit
    z5 = torch.randn(5, 4, 16, 16, generator=g).to(G.dev())
    d5 = hip.decode(z5).sample.clone()
    for i in (0, 3, 4):
        assert torch.equal(hip.decode(z5[i:i + 1]).sample, d5[i:i + 1]), i
    assert torch.equal(hip.decode(z5[1:4]).sample, d5[1:4])
    x5 = (torch.rand(4, 3, 32, 32, generator=g) * 2 - 1).to(G.dev())
    m5 = hip.encode(x5).latent_dist.mode().clone()
    assert torch.equal(hip.encode(x5[2:3]).latent_dist.mode(), m5[2:3])


# (the full-size autoencoder against the oracle: tests/test_gpu_sd_shape_style.py)


def test_error_paths(tiny):
    hip, _ = tiny
    with pytest.raises(Exception, match="multiple of 64"):
        hip.decode(torch.randn(1, 4, 6, 6).to(G.dev()))
    with pytest.raises(ValueError):
        hip.encode(torch.randn(1, 3, 33, 32).to(G.dev()))
    from hedit.vae import AutoencoderKL, TINY_VAE_CONFIG
    fresh = AutoencoderKL(TINY_VAE_CONFIG, device=G.dev())
    with pytest.raises(Exception, match="unloaded"):
        fresh.decode(torch.randn(1, 4, 8, 8).to(G.dev()))


# ---------------------------------------------------------------- decoder input gradient
@pytest.mark.parametrize("B,h,w", [(1, 16, 16), (2, 8, 8), (3, 16, 8)])
def test_decode_vjp_matches_oracle_autograd(tiny, B, h, w):
    """d_z = J^T d_image from the HIP backward pass == torch.autograd.grad through the CPU restatement,
    which is how the reference's style closure obtains it (n-style h_edit.py:183)."""
    hip, om = tiny
    g = torch.Generator().manual_seed(7 * B + h)
    z = torch.randn(B, 4, h, w, generator=g)
    d_img = torch.randn(B, 3, 2 * h, 2 * w, generator=g)
    zz = z.clone().requires_grad_(True)
    (want,) = torch.autograd.grad((om.decode(zz).sample * d_img).sum(), zz)
    got = hip.decode_vjp(z.to(G.dev()), d_img.to(G.dev()))
    G.sync()
    assert got.shape == z.shape
    G.within(G.rel_err(got, want), 4e-2)          # bf16 activations and gradients, fp32 accumulation


def test_decode_is_differentiable_through_autograd(tiny):
    """the facade's decode is an autograd node: loss.backward() / autograd.grad reach the latents the
    way the reference's closure expects, and the forward value is the plain decode."""
    hip, om = tiny
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 4, 8, 8, generator=g)
    target = torch.randn(2, 3, 16, 16, generator=g)

    def loss_of(model, zz, dev):
        img = model.decode(zz / 0.18215).sample
        return torch.linalg.norm(img.float() - target.to(dev))

    zc = z.clone().requires_grad_(True)
    (want,) = torch.autograd.grad(loss_of(om, zc, "cpu"), zc)
    zg = z.to(G.dev()).requires_grad_(True)
    lg = loss_of(hip, zg, G.dev())
    (got,) = torch.autograd.grad(lg, zg)
    G.sync()
    G.within(G.rel_err(got, want), 4e-2)
    with torch.no_grad():
        plain = hip.decode(z.to(G.dev()) / 0.18215).sample
    assert torch.equal(plain, hip.decode(zg.detach() / 0.18215).sample)


def test_decode_vjp_is_linear_and_deterministic(tiny):
    """the VJP is linear in d_image (J^T (a u + v) = a J^T u + J^T v, up to bf16 rounding) and
    bit-reproducible run to run (fixed-order reductions)."""
    hip, _ = tiny
    dev = G.dev()
    g = torch.Generator().manual_seed(11)
    z = torch.randn(1, 4, 16, 16, generator=g).to(dev)
    u = torch.randn(1, 3, 32, 32, generator=g).to(dev)
    v = torch.randn(1, 3, 32, 32, generator=g).to(dev)
    a = hip.decode_vjp(z, u)
    b = hip.decode_vjp(z, v)
    c = hip.decode_vjp(z, 2.0 * u + v)
    G.sync()
    G.within(G.rel_err(c, 2.0 * a + b), 2e-2)
    assert torch.equal(a, hip.decode_vjp(z, u))


def test_decode_vjp_sd15_shape_adjoint_identity():
    """SD-1.x decoder at a 64x64 latent: <J v, u> == <v, J^T u> with u aligned to J v (a random u is
    nearly orthogonal to J v and would test nothing).  J v comes from central differences of the HIP
    decode itself at two step sizes, one defining u and one for the left-hand side, so that the bf16
    rounding noise of u is independent of the one in the product (size-independent property; the oracle
    would take minutes here)."""
    from hedit.vae import AutoencoderKL
    dev = G.dev()
    hip = AutoencoderKL(device=dev)
    hip.init_random(5)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 4, 64, 64, generator=g).to(dev)
    v = torch.randn(1, 4, 64, 64, generator=g).to(dev)

    def jv(eps):
        return (hip.decode(z + eps * v).sample - hip.decode(z - eps * v).sample) / (2 * eps)

    u = jv(0.05)
    jtu = hip.decode_vjp(z, u)
    G.sync()
    assert torch.isfinite(jtu).all()
    lhs = (jv(0.02).double() * u.double()).sum().item()
    rhs = (v.double() * jtu.double()).sum().item()
    assert lhs > 0
    assert abs(lhs - rhs) / lhs < 2e-2      # measured 3e-4; the decoder is visibly nonlinear beyond eps ~ 0.1


def test_kept_tape_backward_equals_one_call_vjp(tiny):
    """autograd path (decode_keep + decode_backward) == hedit_vae_decode_vjp bit for bit, and an
    interleaved second differentiable decode makes the first backward fall back, not misfire."""
    hip, _ = tiny
    dev = G.dev()
    g = torch.Generator().manual_seed(21)
    z1 = torch.randn(2, 4, 8, 8, generator=g).to(dev)
    z2 = torch.randn(2, 4, 8, 8, generator=g).to(dev)
    u = torch.randn(2, 3, 16, 16, generator=g).to(dev)
    a = z1.clone().requires_grad_(True)
    (ga,) = torch.autograd.grad((hip.decode(a).sample * u).sum(), a)
    assert torch.equal(ga, hip.decode_vjp(z1, u))
    a = z1.clone().requires_grad_(True)
    b = z2.clone().requires_grad_(True)
    ia = hip.decode(a).sample
    ib = hip.decode(b).sample              # takes the tape slot
    (gb,) = torch.autograd.grad((ib * u).sum(), b)
    (ga2,) = torch.autograd.grad((ia * u).sum(), a)
    G.sync()
    assert torch.equal(gb, hip.decode_vjp(z2, u))
    assert torch.equal(ga2, ga)
