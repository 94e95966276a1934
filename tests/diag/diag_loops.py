"""Diagnostic (not a test): error growth of the HIP loop vs the oracle loop for several output
scales of the synthetic network and chain lengths."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "h-edit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import gpu as G
from helpers.models import make_pair
from helpers.tiny import PROMPT_PAIRS
from hedit.unet import TINY_CONFIG
from oracle import loops as OL, p2p as OP
from hedit.inversion import p2p_h_edit as HE
from hedit.p2p import ptp_controller_utils as PCU
from hedit.p2p.ptp_utils import register_attention_control

T = 8
for out_scale in (1.0, 0.3, 0.1):
    hip, om, sd = make_pair(TINY_CONFIG, T)
    sd2 = dict(sd)
    sd2["conv_out.weight"] = sd["conv_out.weight"] * out_scale
    sd2["conv_out.bias"] = sd["conv_out.bias"] * out_scale
    hip.unet.load_state_dict(sd2); om.unet.load_state_dict(sd2)
    torch.manual_seed(11)
    w0 = torch.randn(1, 4, 32, 32) * 0.8
    pi = 0
    src, tar, blend, is_replace = PROMPT_PAIRS[pi]
    torch.manual_seed(100)
    zs, wts, noise = OL.ddpm_inversion(om, w0, eta=1.0, prompt=src, cfg_src=1.0, T=T)
    x = torch.randn(4, 4, 32, 32); ctx = torch.randn(4, 77, 64)
    with torch.no_grad():
        want = om.unet(x, torch.tensor(501), encoder_hidden_states=ctx).sample
    got = hip.unet(G.f32(x), 501, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"use_controller": False}).sample
    print(f"scale {out_scale}: single-call rel err {G.rel_err(got, want):.3e}  |eps| {want.abs().mean():.3f} zs max {zs.abs().max():.1f}")
    for after in (1, 2, 4, 8):
        for K in (1, 2):
            bw = ((blend[0],), (blend[1],)); eq = {"words": (blend[1],), "values": (2.0,)}
            hc = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=after, tokenizer=hip.tokenizer, device=hip.device)
            oc = OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=after, tok=om.tokenizer)
            register_attention_control(hip, hc); OP.register(om, oc)
            kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], after_skip_steps=after, is_ddim_inversion=False, weight_reconstruction=0.1, optimization_steps=K)
            with torch.no_grad():
                e_o, r_o = OL.h_edit_p2p_implicit(om, xT=wts[after], zs=zs[:after], controller=oc, **kw)
            e_h, r_h = HE.h_Edit_p2p_implicit(hip, xT=G.f32(wts[after]), zs=G.f32(zs[:after]), controller=hc, **kw)
            print(f"   after={after} K={K}: edit err {G.rel_err(e_h, e_o):.3e} recon err {G.rel_err(r_h, r_o):.3e}  |edit| {e_o.abs().mean():.2f} recon-vs-w0 {G.rel_err(r_h, w0):.3e}")
