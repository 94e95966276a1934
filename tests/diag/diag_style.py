"""Measured errors of the text + style loop (HIP vs oracle) on the tiny synthetic models: the numbers the
tolerances in tests/test_gpu_style.py are derived from."""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "h-edit_amd")):
    sys.path.insert(0, p)
import test_gpu_style as TS
from helpers import gpu as G
from helpers.tiny import PROMPT_PAIRS
from oracle import loops as OL
from hedit.inversion import h_edit as HS
from hedit.engine import HEditEngine

hip, om, inv, enc, enc_g = TS.setup.__wrapped__() if hasattr(TS.setup, "__wrapped__") else TS.setup.__pytest_wrapped__.obj()
g = torch.Generator().manual_seed(5)
cfg = [1.0, 5.0, 7.5]
for trial in range(3):
    e_u, e_cs, e_ct = (torch.randn(1, 4, 32, 32, generator=g) for _ in range(3))
    x = torch.randn(1, 4, 32, 32, generator=g)
    tt = int(om.scheduler.timesteps[3 + trial])
    e_hat = e_u + cfg[1] * (e_cs - e_u); e_tar = e_u + cfg[2] * (e_ct - e_u)
    want = OL._style_step(om, enc, x, e_tar, e_tar - e_hat, tt, 0.5)
    got = HEditEngine(hip).style_step(G.f32(e_u), G.f32(e_cs), G.f32(e_u), G.f32(e_ct), G.f32(x), tt, cfg, enc_g, 0.5)
    print("style step", tt, "rel err of the step", G.rel_err(got - G.f32(x), want - x), "of x", G.rel_err(got, want))
for (pi, skip, K, weight, with_enc) in TS.CASES + [(2, 4, 2, 0.55, False), (2, 4, 1, 0.55, True)]:
    zs, wts = inv[pi]
    after = TS.T - skip
    hc, oc = TS.controllers(hip, om, pi, after)
    prompts = [PROMPT_PAIRS[pi][0], PROMPT_PAIRS[pi][1]]
    kw = dict(eta=1.0, prompts=prompts, cfg_scales=cfg, weight_edit_clip=weight, optimization_steps=K,
              after_skip_steps=after, is_ddim_inversion=False)
    e_o, r_o = OL.h_edit_p2p_implicit_style(om, enc if with_enc else None, wts[after], zs=zs[:after], controller=oc, **kw)
    e_h, r_h = HS.h_Edit_p2p_implicit(hip, enc_g if with_enc else None, xT=G.f32(wts[after]), zs=G.f32(zs[:after]),
                                      controller=hc, prog_bar=False, **kw)
    print((pi, skip, K, weight, with_enc), "edit", round(G.rel_err(e_h, e_o), 4), "recon", round(G.rel_err(r_h, r_o), 4))
