"""HIP side of the SD-1.5-shape loop trajectory comparison: BASELINE configs[1] (h_Edit_p2p_implicit, 50-step schedule, K = 1, P2P
Replace + Reweight + LocalBlend) on the inputs of the committed ORACLE trajectory (tests/golden/t1_sd15_loop_trajectory_g*.npz, made by
tests/golden/make_loop_trajectory.py in the build container), relative L2 distance to the oracle after every step.  Seconds of GPU
time and no host oracle."""
import os

import numpy as np
import torch

from helpers import gpu as G
from helpers.models import make_hip
from helpers.tiny import PROMPT_PAIRS

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")


def fixture_path(gain=0.3):
    return os.path.join(GOLDEN, f"t1_sd15_loop_trajectory_g{gain:g}.npz")


def hip_vs_oracle_trajectory(gain=0.3):
    """-> (timesteps, edited rel L2 per step, reconstruction rel L2 per step, |x_edit| rms of the oracle per step)"""
    from hedit.inversion import p2p_h_edit as HE
    from hedit.p2p import ptp_controller_utils as PCU
    from hedit.p2p.ptp_utils import register_attention_control
    from hedit.unet import SD15_CONFIG
    fx = np.load(fixture_path(gain))
    T = int(fx["zs"].shape[0])
    hip = make_hip(SD15_CONFIG, T, seed=3, out_scale=float(fx["out_scale"]))
    src, tar, blend, is_replace = PROMPT_PAIRS[0]
    bw = ((blend[0],), (blend[1],))
    eq = {"words": (blend[1],), "values": (2.0,)}
    hc = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=T, tokenizer=hip.tokenizer,
                             device=hip.device)
    got = []
    inner = hc.step_callback

    def cb(xt):
        xt = inner(xt)
        got.append(xt.detach().float().cpu().clone())
        return xt
    hc.step_callback = cb
    register_attention_control(hip, hc)
    kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], after_skip_steps=T, is_ddim_inversion=False,
              weight_reconstruction=0.1, optimization_steps=1)
    HE.h_Edit_p2p_implicit(hip, xT=G.f32(torch.from_numpy(fx["xT"])), zs=G.f32(torch.from_numpy(fx["zs"])), controller=hc, prog_bar=False, **kw)
    G.sync()
    want = torch.from_numpy(fx["trace"])
    assert len(got) == want.shape[0] == T
    rel = lambda a, b: float((a - b).norm() / b.norm())          # noqa: E731
    e_edit = [rel(g[1:], w[1:]) for g, w in zip(got, want)]
    e_rec = [rel(g[:1], w[:1]) for g, w in zip(got, want)]
    rms = [float(w[1:].pow(2).mean().sqrt()) for w in want]
    return [int(t) for t in fx["timesteps"]][-T:], e_edit, e_rec, rms
