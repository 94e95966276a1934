"""-m gpu: MasaCtrl mutual self-attention on the HIP path -- the kv_src indirection of
the self-attention kernel and h_Edit_masactrl_implicit -- against the oracle pinned on the reference's own
MasaCtrl classes (tests/test_oracle_masactrl.py, g13)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import gpu as G  # noqa: E402
from helpers.models import make_pair  # noqa: E402
from helpers.tiny import PROMPT_PAIRS  # noqa: E402
from hedit import _lib  # noqa: E402
from hedit.unet import TINY_CONFIG  # noqa: E402

T = 8


@pytest.mark.parametrize("heads,d,N", [(8, 40, 1024), (4, 64, 256), (8, 80, 256), (8, 160, 64)])
def test_self_attention_reads_kv_of_another_row(heads, d, N):
    """row b: softmax(q_b k_src^T) v_src with src = kv_src[b]; combined with qk_src the key index follows kv_src"""
    lib = _lib.lib()
    B, Cc = 4, heads * d
    g = torch.Generator().manual_seed(heads * 1000 + N)
    q, k, v = (torch.randn(B, N, Cc, generator=g) * 0.5 for _ in range(3))
    scale = d ** -0.5 * 1.4426950408889634
    qk = torch.cat([q * scale, k], dim=-1).to(_lib.storage_dtype()).to(G.dev()).contiguous()
    vt = v.to(_lib.storage_dtype()).permute(2, 0, 1).reshape(Cc, B * N).contiguous().to(G.dev())
    out = torch.empty(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    src = torch.tensor([0, 0, 2, 2], dtype=torch.int32, device=G.dev())
    k_view = qk.view(B * N, 2 * Cc)[:, Cc:]
    _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * Cc, C.c_void_p(k_view.data_ptr()), 2 * Cc, _lib.ptr(vt), B * N,
                                     _lib.ptr(out), Cc, B, N, heads, d, None, _lib.ptr(src), None))
    G.sync()
    qb = q.to(_lib.storage_dtype()).float().reshape(B, N, heads, d).transpose(1, 2)
    kb = k.to(_lib.storage_dtype()).float().reshape(B, N, heads, d).transpose(1, 2)[src.cpu().long()]
    vb = v.to(_lib.storage_dtype()).float().reshape(B, N, heads, d).transpose(1, 2)[src.cpu().long()]
    want = (torch.softmax(qb @ kb.transpose(-1, -2) * d ** -0.5, -1) @ vb).transpose(1, 2).reshape(B, N, Cc)
    G.within(G.rel_err(out.float(), want), 1.2e-2)


@pytest.fixture(scope="module")
def setup():
    from oracle import loops as OL
    hip, om, _ = make_pair(TINY_CONFIG, T, out_scale=0.3)
    torch.manual_seed(11)
    w0 = torch.randn(1, 4, 32, 32) * 0.8
    torch.manual_seed(100)
    zs, wts, _ = OL.ddpm_inversion(om, w0, eta=1.0, prompt=PROMPT_PAIRS[0][0], cfg_src=1.0, T=T)
    return hip, om, zs, wts


@pytest.mark.parametrize("skip,K,step,layer", [(4, 1, 1, 2), (4, 2, 0, 0), (0, 1, 2, 4), (4, 1, 99, 0)])
def test_masactrl_loop_matches_oracle(setup, skip, K, step, layer):
    from oracle import loops as OL
    from oracle import masactrl as OM
    from hedit.inversion.masactrl_h_edit import h_Edit_masactrl_implicit
    from hedit.masactrl import MutualSelfAttentionControl, regiter_attention_editor_diffusers
    hip, om, zs, wts = setup
    after = T - skip
    ed_h = MutualSelfAttentionControl(step, layer)
    regiter_attention_editor_diffusers(hip, ed_h)
    ed_o = OM.MutualSelfAttention(step, layer)
    OM.register_editor(om, ed_o)
    assert ed_h.num_att_layers == ed_o.num_att_layers
    prompts = [PROMPT_PAIRS[0][0], PROMPT_PAIRS[0][1]]
    kw = dict(eta=1.0, prompts=prompts, cfg_scales=[1.0, 5.0, 7.5], optimization_steps=K, after_skip_steps=after,
              is_ddim_inversion=False)
    e_o, r_o = OL.h_edit_masactrl_implicit(om, wts[after], zs=zs[:after], **kw)
    e_h, r_h = h_Edit_masactrl_implicit(hip, xT=G.f32(wts[after]), zs=G.f32(zs[:after]), prog_bar=False, **kw)
    G.sync()
    assert e_h.shape == (1, 4, 32, 32) and torch.isfinite(e_h).all()
    tol_edit, tol_recon = (8e-2, 1e-2) if after <= 4 else (1.8e-1, 5e-2)
    G.within(G.rel_err(r_h, r_o), tol_recon)
    G.within(G.rel_err(e_h, e_o), tol_edit)
    assert ed_h.cur_step == ed_o.cur_step == after * K


def test_masactrl_changes_the_edit(setup):
    from hedit.inversion.masactrl_h_edit import h_Edit_masactrl_implicit
    from hedit.masactrl import MutualSelfAttentionControl, regiter_attention_editor_diffusers
    hip, _, zs, wts = setup
    outs = []
    for step in (0, 99):
        regiter_attention_editor_diffusers(hip, MutualSelfAttentionControl(step, 0))
        e, _ = h_Edit_masactrl_implicit(hip, xT=G.f32(wts[4]), eta=1.0, prompts=[PROMPT_PAIRS[0][0], PROMPT_PAIRS[0][1]],
                                        cfg_scales=[1.0, 5.0, 7.5], zs=G.f32(zs[:4]), after_skip_steps=4, is_ddim_inversion=False)
        outs.append(e)
    G.sync()
    assert G.rel_err(outs[0], outs[1]) > 1e-2


def test_loop_needs_a_registered_editor():
    from hedit.inversion.masactrl_h_edit import h_Edit_masactrl_implicit
    hip, _, _ = make_pair(TINY_CONFIG, T)
    with pytest.raises(RuntimeError):
        h_Edit_masactrl_implicit(hip, xT=torch.zeros(1, 4, 32, 32), prompts=["a", "b"], cfg_scales=[1, 2, 3])
