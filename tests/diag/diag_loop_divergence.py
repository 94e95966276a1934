"""Diagnostic (not a test): HIP vs the fp32 oracle on the EDITED branch, per sampler step, over the last N steps of BASELINE
configs[1] at SD-1.5 shape (50-step schedule, K = 1, P2P Replace + Reweight + LocalBlend), same weights, same inversion
outputs (the oracle's).  VERDICT round 4, "do this" 3b.  Prints one line per step: relative L2 distance of the edited
latent and of the reconstruction-branch latent after that step; the result is kept as profiles/r05_loop_divergence.txt.

    python tests/diag/diag_loop_divergence.py [steps=12] [out_scale=1.0] [emu] [first]

out_scale damps the synthetic network's output layer (a random-weight eps-network at full gain is not contractive: the chain
amplifies any perturbation of eps, which says nothing about kernels -- tests/helpers/models.py).  `emu`: the oracle is run a
second time with bf16 STORAGE emulated (weights, leaf-module outputs and residual-stream sums rounded to bf16, arithmetic
fp32: the emulation of test_sd15_unet_forward_full_size) and its divergence from the fp32 oracle is printed beside the HIP
path's -- what part of a chain's divergence is the storage format.  `first`: the FIRST `steps` steps of the schedule (t = 981
downwards, stopped early) instead of the last ones.

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
EMU = "emu" in sys.argv[3:]
FIRST = "first" in sys.argv[3:]
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
trace = {"hip": [], "oracle": [], "emu": []}


class _Stop(Exception):
    pass


def record(ctrl, key, limit):
    inner = ctrl.step_callback

    def cb(xt):
        xt = inner(xt)
        trace[key].append(xt.detach().float().cpu().clone())
        if len(trace[key]) >= limit:
            raise _Stop
        return xt
    ctrl.step_callback = cb


after = T if FIRST else steps            # FIRST: the whole schedule is entered at t = 981 and left after `steps` steps
mk_o = lambda: OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=after, tok=om.tokenizer)   # noqa: E731
hc = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=after, tokenizer=hip.tokenizer,
                         device=hip.device)
kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], after_skip_steps=after, is_ddim_inversion=False,
          weight_reconstruction=0.1, optimization_steps=1)


def run_oracle(key):
    oc = mk_o()
    record(oc, key, steps)
    OP.register(om, oc)
    t0 = time.time()
    try:
        with torch.no_grad():
            OL.h_edit_p2p_implicit(om, xT=wts[after], zs=zs[:after], controller=oc, **kw)
    except _Stop:
        pass
    return time.time() - t0


t_or = run_oracle("oracle")
record(hc, "hip", steps)
register_attention_control(hip, hc)
try:
    HE.h_Edit_p2p_implicit(hip, xT=G.f32(wts[after]), zs=G.f32(zs[:after]), controller=hc, prog_bar=False, **kw)
except _Stop:
    pass
G.sync()
if EMU:
    from oracle import sd_unet as OSU
    rnd = lambda t: t.to(torch.bfloat16).float()          # noqa: E731
    with torch.no_grad():
        for p_ in om.unet.parameters():
            p_.copy_(rnd(p_))
    hooks = [m.register_forward_hook(lambda m_, i_, o_: rnd(o_) if isinstance(o_, torch.Tensor) else o_)
             for m in om.unet.modules() if len(list(m.children())) == 0]
    OSU.RESID_STORE = rnd
    run_oracle("emu")
    OSU.RESID_STORE = None
    for h_ in hooks:
        h_.remove()
ts = [int(v) for v in hip.scheduler.timesteps][-after:][:steps]
print(f"# HIP vs fp32 oracle, h_Edit_p2p_implicit, SD-1.5 shape, {'first' if FIRST else 'last'} {steps} of {T} steps (t = {ts[0]} .. {ts[-1]}), K = 1, "
      f"output gain {out_scale}")
print(f"# oracle loop {t_or:.0f} s on the host; rows: [x_orig, x_edit] after the step's LocalBlend" +
      ("; emu = the oracle with bf16 storage emulated (arithmetic fp32)" if EMU else ""))
print("# step     t   edited rel.L2   reconstruction rel.L2   |x_edit| rms" + ("    emu: edited   reconstruction" if EMU else ""))
rel = lambda a, b: float((a - b).norm() / b.norm())          # noqa: E731
for i, (h, o) in enumerate(zip(trace["hip"], trace["oracle"])):
    line = f"  {i + 1:4d}  {ts[i]:4d}     {rel(h[1:], o[1:]):.3e}        {rel(h[:1], o[:1]):.3e}            {float(o[1:].pow(2).mean().sqrt()):.3f}"
    if EMU:
        e = trace["emu"][i]
        line += f"       {rel(e[1:], o[1:]):.3e}     {rel(e[:1], o[:1]):.3e}"
    print(line)
r_h, r_o = trace["hip"][-1][:1], trace["oracle"][-1][:1]
if not FIRST:
    print(f"# reconstruction vs the inverted latent: {rel(r_h, w0):.3e} (HIP on the ORACLE's inversion outputs), {rel(r_o, w0):.3e} (oracle); with its own "
          f"inversion the HIP loop reconstructs exactly (bench.py recon_rel_err 0.0)")
