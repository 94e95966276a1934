"""The oracle's text + style loop against vectors produced by RUNNING the reference's
text-guided-n-style/inversion/h_edit.py (tests/golden/make_golden.py --only-style, g9)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.tiny import PROMPT_PAIRS, TinyStyleEncoder, TinyVae, make_tiny_model  # noqa: E402
from oracle import loops, p2p  # noqa: E402

# the toy trajectories amplify rounding differences (see test_oracle_golden.py): same thread count as
# the generator so the CPU reductions associate identically
torch.set_num_threads(4)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(G, "g9_style.json")))


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(G, "g9_style.npz"))


@pytest.mark.parametrize("case", META["cases"], ids=[c["name"] for c in META["cases"]])
def test_style_loop_matches_reference(vec, case):
    T = META["T"]
    model = make_tiny_model(T)
    model.vae = TinyVae()
    pair = PROMPT_PAIRS[case["pair"]]
    if not case["blend"]:
        pair = pair[:2] + (None, pair[3])
    after = T - case["skip"]
    zs = torch.from_numpy(vec[f"{case['name']}_zs"])
    wts = torch.from_numpy(vec[f"{case['name']}_wts"])
    src, tar, blend, is_replace = pair
    bw = ((blend[0],), (blend[1],)) if blend else None
    eq = {"words": (blend[1],), "values": (2.0,)} if blend else None
    ctrl = p2p.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=after,
                               tok=model.tokenizer)
    p2p.register(model, ctrl)
    enc = TinyStyleEncoder() if case["with_encoder"] else None
    edit, recon = loops.h_edit_p2p_implicit_style(model, enc, wts[after], eta=1.0, prompts=[pair[0], pair[1]],
                                                  cfg_scales=[1.0, 5.0, 7.5], zs=zs[:after], controller=ctrl,
                                                  weight_edit_clip=case["weight"], optimization_steps=case["K"],
                                                  after_skip_steps=after)
    want_e = torch.from_numpy(vec[f"{case['name']}_edit"])
    want_r = torch.from_numpy(vec[f"{case['name']}_recon"])
    assert torch.allclose(recon, want_r, atol=2e-5, rtol=1e-5)
    assert torch.allclose(edit, want_e, atol=1e-4, rtol=1e-4)
    assert ctrl.cur_step == case["cur_step"]


def test_style_changes_the_edit(vec):
    """the style term is not a no-op in the fixtures: with and without the encoder differ."""
    a, b = vec["style_k1_edit"], vec["style_noenc_edit"]
    assert np.abs(a - b).max() > 1e-2
