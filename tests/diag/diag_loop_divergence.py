"""Diagnostic (not a test): HIP vs the fp32 oracle on the EDITED branch, per sampler step, over the last N steps of BASELINE
configs[1] at SD-1.5 shape (50-step schedule, K = 1, P2P Replace + Reweight + LocalBlend), same weights, same inversion
outputs (the oracle's).  VERDICT round 4, "do this" 3b.  Prints one line per step: relative L2 distance of the edited
latent and of the reconstruction-branch latent after that step; the result is kept as profiles/r05_loop_divergence.txt.

    python tests/diag/diag_loop_divergence.py [steps=12] [out_scale=1.0]

(The oracle runs on the host: about 3 s per sample-forward, 9 per step.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "h-edit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from helpers import gpu as G  # noqa: E402
from helpers.models import make_pair  # noqa: E402
from helpers.tiny import PROMPT_PAIRS  # noqa: E402
from hedit.unet import SD15_CONFIG  # noqa: E402
from oracle import loops as OL, p2p as OP  # noqa: E402
from hedit.inversion import p2p_h_edit as HE  # noqa: E402
from hedit.p2p import ptp_controller_utils as PCU  # noqa: E402
from hedit.p2p.ptp_utils import register_attention_control  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
out_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
T = 50
hip, om, _ = make_pair(SD15_CONFIG, T, seed=3, out_scale=out_scale)
src, tar, blend, is_replace = PROMPT_PAIRS[0]
torch.manual_seed(11)
w0 = torch.randn(1, 4, 64, 64) * 0.8
t0 = time.time()
with torch.no_grad():
    torch.manual_seed(100)
    zs, wts, _ = OL.ddpm_inversion(om, w0, eta=1.0, prompt=src, cfg_src=1.0, T=T)
print(f"# oracle DDPM inversion, {T} steps: {time.time() - t0:.0f} s", flush=True)
bw = ((blend[0],), (blend[1],))
eq = {"words": (blend[1],), "values": (2.0,)}
hc = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=steps, tokenizer=hip.tokenizer,
                         device=hip.device)
oc = OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=steps, tok=om.tokenizer)
trace = {"hip": [], "oracle": []}


def record(ctrl, key):
    inner = ctrl.step_callback

    def cb(xt):
        xt = inner(xt)
        trace[key].append(xt.detach().float().cpu().clone())
        return xt
    ctrl.step_callback = cb


record(hc, "hip")
record(oc, "oracle")
register_attention_control(hip, hc)
OP.register(om, oc)
kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], after_skip_steps=steps, is_ddim_inversion=False,
          weight_reconstruction=0.1, optimization_steps=1)
t0 = time.time()
with torch.no_grad():
    e_o, r_o = OL.h_edit_p2p_implicit(om, xT=wts[steps], zs=zs[:steps], controller=oc, **kw)
t_or = time.time() - t0
e_h, r_h = HE.h_Edit_p2p_implicit(hip, xT=G.f32(wts[steps]), zs=G.f32(zs[:steps]), controller=hc, prog_bar=False, **kw)
G.sync()
ts = [int(v) for v in hip.scheduler.timesteps][-steps:]
print(f"# HIP vs fp32 oracle, h_Edit_p2p_implicit, SD-1.5 shape, last {steps} of {T} steps (t = {ts[0]} .. {ts[-1]}), K = 1, output gain {out_scale}")
print(f"# oracle loop {t_or:.0f} s on the host; rows: [x_orig, x_edit] after the step's LocalBlend")
print("# step     t   edited rel.L2   reconstruction rel.L2   |x_edit| rms")
rel = lambda a, b: float((a - b).norm() / b.norm())          # noqa: E731
for i, (h, o) in enumerate(zip(trace["hip"], trace["oracle"])):
    print(f"  {i + 1:4d}  {ts[i]:4d}     {rel(h[1:], o[1:]):.3e}        {rel(h[:1], o[:1]):.3e}            {float(o[1:].pow(2).mean().sqrt()):.3f}")
print(f"# final: edited {G.rel_err(e_h, e_o):.3e}, reconstruction {G.rel_err(r_h, r_o):.3e}; reconstruction vs the inverted latent "
      f"{G.rel_err(r_h, w0):.3e} (HIP), {rel(r_o, w0):.3e} (oracle)")
