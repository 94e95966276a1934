"""-m gpu: Plug-and-Play injection on the HIP path: q/k injection restricted to the decoder
blocks, ResNet feature injection, and h_Edit_PnP_implicit -- against vectors produced by the REFERENCE's hooks and loop
running on the oracle UNet (g14; tests/test_oracle_pnp.py pins the oracle's own restatement on the same vectors)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import gpu as G  # noqa: E402
from helpers.tiny import PROMPT_PAIRS, TINY4_CONFIG  # noqa: E402

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GD, "g14_pnp.json")))
T = 4


def hip_model():
    from helpers.models import make_pair
    hip, _, _ = make_pair(TINY4_CONFIG, T, out_scale=0.3)        # same seed / text encoder as make_oracle_sd_model
    return hip


@pytest.mark.parametrize("case", META, ids=[c["name"] for c in META])
def test_pnp_loop_matches_reference_vectors(case):
    from hedit.inversion.pnp_h_edit import h_Edit_PnP_implicit
    from hedit.plug_n_play import register_attention_control_efficient, register_conv_control_efficient
    vec = np.load(os.path.join(GD, "g14_pnp.npz"))
    hip = hip_model()
    register_attention_control_efficient(hip, case["qk"])
    register_conv_control_efficient(hip, case["conv"])
    zs = torch.from_numpy(vec[f"{case['name']}_zs"])
    wts = torch.from_numpy(vec[f"{case['name']}_wts"])
    edit, recon = h_Edit_PnP_implicit(hip, xT=G.f32(wts[T]), eta=1.0, prompts=[PROMPT_PAIRS[0][0], PROMPT_PAIRS[0][1]],
                                      cfg_scales=[1.0, 5.0, 7.5], prog_bar=False, zs=G.f32(zs[:T]), optimization_steps=case["K"],
                                      after_skip_steps=T, is_ddim_inversion=False)
    G.sync()
    assert edit.shape == (1, 4, 64, 64) and torch.isfinite(edit).all()
    G.within(G.rel_err(recon, torch.from_numpy(vec[f"{case['name']}_recon"])), 3e-2)      # measured 1.1e-2 (4 steps of 250 timesteps each)
    G.within(G.rel_err(edit, torch.from_numpy(vec[f"{case['name']}_edit"])), 8e-2)     # 4-step chain of bf16 eps evaluations


def test_injection_changes_the_edit_and_only_for_two_rows():
    from hedit.plug_n_play import register_attention_control_efficient, register_conv_control_efficient, register_time
    hip = hip_model()
    dev = G.dev()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 4, 64, 64, generator=g).to(dev)
    ctx = torch.randn(2, 77, 64, generator=g).to(dev)
    plain = hip.unet(x, 501, encoder_hidden_states=ctx).sample
    register_attention_control_efficient(hip, [501])
    register_conv_control_efficient(hip, [501])
    register_time(hip, 501)
    inj = hip.unet(x, 501, encoder_hidden_states=ctx).sample
    register_time(hip, 251)                      # outside the schedule
    off = hip.unet(x, 501, encoder_hidden_states=ctx).sample
    register_time(hip, 501)
    four = hip.unet(torch.cat([x, x]), 501, encoder_hidden_states=torch.cat([ctx, ctx])).sample   # B // 2 == 2: silent
    G.sync()
    assert torch.equal(inj[:1], plain[:1])                       # the source row is untouched
    assert G.rel_err(inj[1:], plain[1:]) > 1e-2                  # the target row is not
    assert torch.equal(off, plain)
    G.within(G.rel_err(four[:2], plain), 1e-2)


def test_lock_step_pnp_equals_the_single_image_runs():
    """engine.run_pnp with two images in lock-step (2n-row injected pass: row n + i takes row i's q, k / features) gives
    each image exactly what the one-image run gives it (batch-invariant kernels, same arithmetic)."""
    from hedit.engine import HEditEngine
    from hedit.plug_n_play import register_attention_control_efficient, register_conv_control_efficient
    vec = np.load(os.path.join(GD, "g14_pnp.npz"))
    case = META[-1]
    hip = hip_model()
    register_attention_control_efficient(hip, case["qk"])
    register_conv_control_efficient(hip, case["conv"])
    eng = HEditEngine(hip)
    zs = G.f32(torch.from_numpy(vec[f"{case['name']}_zs"])[:T]).reshape(T, 4, 64, 64)
    xT = G.f32(torch.from_numpy(vec[f"{case['name']}_wts"])[T]).reshape(1, 4, 64, 64)
    g = torch.Generator().manual_seed(9)
    xT2 = torch.cat([xT, G.f32(torch.randn(1, 4, 64, 64, generator=g))])
    zs2 = torch.stack([zs, G.f32(torch.randn(T, 4, 64, 64, generator=g))], 1)
    pairs = [[PROMPT_PAIRS[0][0], PROMPT_PAIRS[0][1]], [PROMPT_PAIRS[1][0], PROMPT_PAIRS[1][1]]]
    kw = dict(cfg_scales=[1.0, 5.0, 7.5], eta=1.0, K=case["K"], after_skip_steps=T, ddim_inv=False)
    e_all, r_all = eng.run_pnp(xT2, zs2, pairs, **kw)
    for i in range(2):
        e1, r1 = eng.run_pnp(xT2[i:i + 1], zs2[:, i:i + 1].contiguous(), [pairs[i]], **kw)
        G.sync()
        assert torch.equal(e_all[i:i + 1], e1) and torch.equal(r_all[i:i + 1], r1)
