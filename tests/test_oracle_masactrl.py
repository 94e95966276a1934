"""The oracle's MasaCtrl editor (oracle/masactrl.py) and loop (oracle/loops.py::h_edit_masactrl_implicit) against
vectors produced by RUNNING the reference's MutualSelfAttentionControl, its registration and
h_Edit_masactrl_implicit on the toy UNet (tests/golden/make_golden.py::gen_masactrl, g13)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.tiny import PROMPT_PAIRS, make_tiny_masa_model  # noqa: E402
from oracle import loops, masactrl  # noqa: E402

torch.set_num_threads(4)     # as the generator (see test_oracle_golden.py)
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(G, "g13_masactrl.json")))


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(G, "g13_masactrl.npz"))


@pytest.mark.parametrize("case", META, ids=[c["name"] for c in META])
def test_masactrl_loop_matches_reference(vec, case):
    T = 10
    model = make_tiny_masa_model(T)
    ed = masactrl.MutualSelfAttention(case["start_step"], case["start_layer"])
    masactrl.register_editor(model, ed)
    assert ed.num_att_layers == case["num_att_layers"]
    after = T - case["skip"]
    zs = torch.from_numpy(vec[f"{case['name']}_zs"])
    wts = torch.from_numpy(vec[f"{case['name']}_wts"])
    pair = PROMPT_PAIRS[case["pair"]]
    edit, recon = loops.h_edit_masactrl_implicit(model, wts[after], eta=1.0, prompts=[pair[0], pair[1]],
                                                 cfg_scales=[1.0, 5.0, 7.5], zs=zs[:after], optimization_steps=case["K"],
                                                 after_skip_steps=after, is_ddim_inversion=case["ddim"])
    assert torch.allclose(recon, torch.from_numpy(vec[f"{case['name']}_recon"]), atol=2e-5, rtol=1e-5)
    assert torch.allclose(edit, torch.from_numpy(vec[f"{case['name']}_edit"]), atol=1e-4, rtol=1e-4)
    assert ed.cur_step == case["cur_step"]


def test_mutual_attention_changes_the_edit(vec):
    """the fixtures exercise the editor: start_step beyond the run (editor never active) differs from an active one"""
    assert np.abs(vec["masa_k1_edit"] - vec["masa_off_edit"]).max() > 1e-2
