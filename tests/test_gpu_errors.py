"""-m gpu: error behaviour of the newer C-ABI entry points -- every misuse comes back as a return code + message
(raised as hedit._lib.HipError by the host wrappers), never a crash or a silent fallback."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import gpu as G  # noqa: E402
from hedit import _lib  # noqa: E402


def test_vae_vjp_workspace_too_small_and_unloaded_params():
    from hedit.vae import AutoencoderKL, TINY_VAE_CONFIG
    vae = AutoencoderKL(TINY_VAE_CONFIG, device=G.dev())
    lib = _lib.lib()
    z = torch.zeros(1, 4, 8, 8, device=G.dev())
    u = torch.zeros(1, 3, 16, 16, device=G.dev())
    dz = torch.empty_like(z)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=G.dev())
    rc = lib.hedit_vae_decode_vjp(vae._h, _lib.ptr(z), _lib.ptr(u), 1, 8, 8, _lib.ptr(dz), None, _lib.ptr(ws), ws.numel(), None)
    assert rc != 0 and b"unloaded" in lib.hedit_last_error()
    vae.init_random(0)
    small = torch.empty(4096, dtype=torch.uint8, device=G.dev())
    rc = lib.hedit_vae_decode_vjp(vae._h, _lib.ptr(z), _lib.ptr(u), 1, 8, 8, _lib.ptr(dz), None, _lib.ptr(small), small.numel(), None)
    assert rc != 0 and b"workspace too small" in lib.hedit_last_error()
    with pytest.raises(ValueError):
        vae.decode_vjp(z, torch.zeros(1, 3, 8, 8, device=G.dev()))            # d_image of the wrong size


def test_decode_backward_needs_a_kept_forward_and_its_workspace():
    from hedit.vae import AutoencoderKL, TINY_VAE_CONFIG
    vae = AutoencoderKL(TINY_VAE_CONFIG, device=G.dev())
    vae.init_random(0)
    lib = _lib.lib()
    u = torch.zeros(1, 3, 16, 16, device=G.dev())
    dz = torch.empty(1, 4, 8, 8, device=G.dev())
    ws = torch.empty(1 << 24, dtype=torch.uint8, device=G.dev())
    rc = lib.hedit_vae_decode_backward(vae._h, _lib.ptr(u), _lib.ptr(dz), _lib.ptr(ws), None)
    assert rc != 0 and b"no forward is being kept" in lib.hedit_last_error()
    z = torch.randn(1, 4, 8, 8, device=G.dev())
    img, _ = vae._decode_keep(z)
    other = torch.empty(1 << 24, dtype=torch.uint8, device=G.dev())
    rc = lib.hedit_vae_decode_backward(vae._h, _lib.ptr(u), _lib.ptr(dz), _lib.ptr(other), None)
    assert rc != 0 and b"not the workspace" in lib.hedit_last_error()
    got = vae._decode_backward(z, u)                      # the tape is still intact after the refused call
    assert torch.isfinite(got).all()
    rc = lib.hedit_vae_decode_backward(vae._h, _lib.ptr(u), _lib.ptr(dz), _lib.ptr(vae._ws_tape), None)
    assert rc != 0                                        # one backward per forward


def test_ddpm_config_and_state_errors():
    from hedit.diffusion import Model, TINY_DDPM_CONFIG
    with pytest.raises(_lib.HipError):
        Model(dict(TINY_DDPM_CONFIG, ch=48), device=G.dev())                 # ch must be a multiple of 64
    with pytest.raises(_lib.HipError):
        Model(dict(TINY_DDPM_CONFIG, image_size=20, attn_resolutions=(20,)), device=G.dev())   # 400 tokens: not a multiple of 64
    with pytest.raises(NotImplementedError):
        Model(dict(TINY_DDPM_CONFIG, resamp_with_conv=False), device=G.dev())
    m = Model(TINY_DDPM_CONFIG, device=G.dev())
    with pytest.raises(_lib.HipError, match="unloaded"):
        m(torch.zeros(1, 3, 32, 32, device=G.dev()), 1.0)
    sd = m.init_random(0)
    with pytest.raises(KeyError):
        m.load_state_dict({k: v for k, v in sd.items() if k != "conv_in.bias"})
    bad = dict(sd)
    bad["conv_in.weight"] = torch.zeros(64, 3, 1, 1)
    with pytest.raises(ValueError):
        m.load_state_dict(bad)
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 3, 16, 16, device=G.dev()), 1.0)                    # the reference asserts the resolution too


def test_masactrl_and_pnp_argument_errors():
    from helpers.models import make_pair
    from hedit.masactrl import MutualSelfAttentionControl
    from hedit.plug_n_play import register_attention_control_efficient
    from hedit.unet import TINY_CONFIG
    with pytest.raises(NotImplementedError):
        MutualSelfAttentionControl(layer_idx=[3, 7])                         # non-contiguous layer set
    hip, _, _ = make_pair(TINY_CONFIG, 4)
    ed = MutualSelfAttentionControl(0, 0)
    with pytest.raises(ValueError):
        ed._plan(hip.unet, 3, 32, 32, True)                                  # not 4 rows per image
    register_attention_control_efficient(hip, [501])
    with pytest.raises(NotImplementedError):                                 # three-level toy: no up_blocks[3]
        from hedit.plug_n_play import register_time
        register_time(hip, 501)
        hip.unet(torch.zeros(2, 4, 32, 32, device=G.dev()), 501, encoder_hidden_states=torch.zeros(2, 77, 64, device=G.dev()))
