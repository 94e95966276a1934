"""CPU tests of the product's host logic (no GPU, no oracle in the product path): P2P tables vs
the reference-generated golden vectors, the folded (A_s, bvec_s) mixing tables vs the reference
controller's outputs, scheduler coefficients, and that the C-ABI library loads and exports every
symbol declared in include/hedit.h."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from helpers.tiny import PROMPT_PAIRS, WordTokenizer, ddim_tables, hash_probs
from hedit import _lib
from hedit.engine import Schedule
from hedit.p2p import ptp_controller_utils as PCU
from hedit.p2p import ptp_utils as PU
from hedit.scheduler import DDIMScheduler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _json(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def make(pair, num_steps, eq_val, tok):
    src, tar, blend, is_replace = pair
    bw = ((blend[0],), (blend[1],)) if blend else None
    eq = {"words": (blend[1],), "values": (eq_val,)} if blend else None
    return PCU.make_controller(prompts=[src, tar], is_replace_controller=is_replace, cross_replace_steps=0.4,
                               self_replace_steps=0.35, blend_word=bw, equilizer_params=eq,
                               num_steps=num_steps, tokenizer=tok, device=None)


@pytest.mark.parametrize("pi", range(len(PROMPT_PAIRS)))
def test_tables_match_reference(golden_dir, pi):
    g = _npz(golden_dir, "g4_controller.npz")
    info = _json(golden_dir, "g4_controller.json")["pairs"][pi]
    tok = WordTokenizer(split_long_words_at=6)
    c = make(PROMPT_PAIRS[pi], 50, info["eq_val"], tok)
    assert type(c).__name__ == info["class"]
    base = c.prev_controller if getattr(c, "prev_controller", None) is not None else c
    assert type(base).__name__ == info["base_class"]
    assert np.array_equal(base.mapper.numpy(), g[f"p{pi}_mapper"])
    if f"p{pi}_alphas" in g:
        assert np.array_equal(base.alphas.numpy(), g[f"p{pi}_alphas"])
    if f"p{pi}_equalizer" in g:
        assert np.array_equal(c.equalizer.numpy(), g[f"p{pi}_equalizer"])
    assert np.array_equal(c.cross_replace_alpha.numpy(), g[f"p{pi}_cross_replace_alpha"])
    assert list(c.num_self_replace) == info["num_self_replace"]
    if c.local_blend is not None:
        assert np.array_equal(c.local_blend.alpha_layers.numpy(), g[f"p{pi}_lb_alpha_layers"])
        assert c.local_blend.start_blend == info["lb_start_blend"]
    for key, want in info["word_inds"].items():
        text, w = key.split("|")
        assert [int(v) for v in PU.get_word_inds(text, w, tok)] == want


@pytest.mark.parametrize("pi", range(len(PROMPT_PAIRS)))
def test_mix_tables_reproduce_reference_edit(golden_dir, pi):
    """P_src . A_s + bvec_s * P_tar must equal what the reference controller wrote into the
    target-conditional rows (cross layers of the golden pass)."""
    g = _npz(golden_dir, "g4_controller.npz")
    info = _json(golden_dir, "g4_controller.json")["pairs"][pi]
    tok = WordTokenizer(split_long_words_at=6)
    c = make(PROMPT_PAIRS[pi], 50, info["eq_val"], tok)
    A, b = c._mix_tables()
    assert A.shape == (51, 77, 77) and b.shape == (51, 77)
    for cur_step in (0, 16, 17, 19, 20, 49):
        for li, (is_cross, place, n) in enumerate(info["layers"]):
            if not is_cross:
                continue
            probs = hash_probs((4, n, 77), 100000 + pi * 1000 + cur_step * 10 + li)
            src, tar = probs[2], probs[3]
            new = src @ A[cur_step] + b[cur_step] * tar
            want = torch.from_numpy(g[f"p{pi}_s{cur_step}_l{li}_tar"])[0]
            assert (new - want).abs().max().item() <= 2e-6


def test_replace_requires_equal_word_count():
    tok = WordTokenizer()
    with pytest.raises(ValueError):
        make(("a b c", "a b c d", None, True), 10, 2.0, tok)


@pytest.mark.parametrize("T", [10, 20, 50])
def test_scheduler_and_coefficients(golden_dir, T):
    g = _json(golden_dir, "g1_scheduler.json")[str(T)]
    s = DDIMScheduler()
    s.set_timesteps(T)
    assert [int(t) for t in s.timesteps] == g["timesteps"]
    ref = ddim_tables(T)
    assert torch.equal(s.alphas_cumprod, ref.alphas_cumprod)
    assert abs(float(s.final_alpha_cumprod) - g["final_alpha_cumprod"]) < 1e-9
    S = Schedule(s)
    for row in g["rows"]:
        t, tt = row["t"], row["tt"]
        assert abs(float(S.variance(t)) - row["variance"]) <= 1e-7 + 1e-6 * abs(row["variance"])
        for eta in (0.0, 1.0):
            for ddim in (False, True):
                want = row[f"coeff_eta{int(eta)}_ddim{int(ddim)}"]
                assert abs(float(S.full_coeff(t, tt, eta, ddim)) - want) <= 1e-6


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hedit.h")).read()
    declared = sorted(set(re.findall(r"\b(hedit_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    assert os.path.exists(_lib.LIB_PATH), "build libhedit_hip.so first: python h-edit_amd/build.py"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hedit.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "ctypes signature table out of sync with include/hedit.h"
    assert _lib.lib().hedit_version() >= 1


def test_half_storage_library_exports_the_same_abi():
    """libhedit_hip_f16.so (build.py --f16: the same sources with -DHEDIT_STORE_F16) exports every declared symbol, says what it
    is, and a process that asks for one storage format is not handed the other library."""
    import subprocess
    import sys
    f16 = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhedit_hip_f16.so")
    assert os.path.exists(f16), "build it: python h-edit_amd/build.py --f16"
    lib = ctypes.CDLL(f16)
    for name in _lib.EXPORTS:
        assert hasattr(lib, name), f"{name} not exported by the half-storage build"
    assert lib.hedit_storage_is_f16() == 1 and _lib.lib().hedit_storage_is_f16() == 0
    code = ("import sys; sys.path.insert(0, %r); from hedit import _lib; import torch; "
            "assert _lib.LIB_PATH.endswith('libhedit_hip_f16.so') and _lib.lib().hedit_storage_is_f16() == 1 "
            "and _lib.storage_dtype() == torch.float16; print('ok')") % os.path.join(ROOT, "h-edit_amd")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HEDIT_STORAGE="f16"), capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HEDIT_STORAGE="fp8"), capture_output=True, text=True)
    assert r.returncode != 0 and "HEDIT_STORAGE" in r.stderr


def test_no_fallback_without_library(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhedit_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "h-edit_amd", "hedit")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports oracle"


def test_sample_xts_and_slerp_helpers():
    """small host helpers with the reference's names (ddpm_inversion.py:5-52, inversion_utils.py:142-160)"""
    import types
    from hedit.inversion.ddpm_inversion import sample_xts_from_x0
    from hedit.inversion.inversion_utils import slerp, slerp_tensor
    from hedit.scheduler import DDIMScheduler
    sch = DDIMScheduler()
    sch.set_timesteps(10)
    model = types.SimpleNamespace(scheduler=sch, device=torch.device("cpu"))
    x0 = torch.randn(4, 8, 8)
    torch.manual_seed(3)
    xts, nz = sample_xts_from_x0(model, x0, num_inference_steps=10)
    assert xts.shape == nz.shape == (11, 4, 8, 8) and torch.equal(xts[0], x0) and not nz[0].any()
    ts = [int(t) for t in sch.timesteps]
    for pos, t in enumerate(ts):
        idx = 10 - pos
        ab = float(sch.alphas_cumprod[t])
        assert torch.allclose(xts[idx], x0 * ab ** 0.5 + nz[idx] * (1 - ab) ** 0.5, atol=1e-6)
    assert ts[-1] < ts[0] and xts[10].std() > xts[1].std() * 0.5      # idx grows with the noise level
    a, b = torch.randn(2, 3, 4, 4), torch.randn(2, 3, 4, 4)
    assert torch.allclose(slerp_tensor(0.0, a, b), a, atol=1e-5) and torch.allclose(slerp_tensor(1.0, a, b), b, atol=1e-5)
    mid = slerp(0.5, a.flatten(1), b.flatten(1))
    assert mid.shape == (2, 48)


def test_word_tokenizer_stable_ids_do_not_depend_on_prompt_order():
    """The synthetic-weights pipelines (drivers with --random_init, HEditPipeline.from_random) give a word the same id
    whatever was tokenised before it, so a prompt embeds identically alone and inside a lock-step batch; the default
    (ids in order of first sight) is what the golden vectors were generated with and stays."""
    from hedit.text import WordTokenizer
    a, b = WordTokenizer(stable_ids=True), WordTokenizer(stable_ids=True)
    p1, p2 = "a cat sitting on a bench", "a blue car on a road"
    ia1, ia2 = a.encode(p1), a.encode(p2)
    ib2, ib1 = b.encode(p2), b.encode(p1)
    assert ia1 == ib1 and ia2 == ib2
    assert a.decode(ia1[1:-1]) == p1.replace(" ", "")
    assert len(set(ia1[1:-1])) == len(set(p1.split(" ")))
    c, d = WordTokenizer(), WordTokenizer()
    assert c.encode(p1)[1:4] == [1, 2, 3] and d.encode(p2)[1:4] == [1, 2, 3]


def test_word_tokenizer_stable_ids_survive_collisions():
    """A real prompt set collides in the ~49 k id range (birthday bound: ~10 % at 100 words): the later word is re-hashed
    with a counter and reported, nothing raises in the middle of a run; `prescan` with the run's vocabulary makes the
    resolution independent of the order the prompts are tokenised in."""
    import warnings
    from hedit.text import WordTokenizer, prescan_prompts

    class Small(WordTokenizer):           # 40 id slots: 30 words must collide
        bos_token_id, eos_token_id = 41, 42

    words = [f"w{i}" for i in range(30)]
    prompts = [" ".join(words[i:i + 5]) for i in range(0, 30, 5)]
    a = Small(stable_ids=True)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ids = [a.encode(p)[1:-1] for p in prompts]
    flat = [i for row in ids for i in row]
    assert len(set(flat)) == 30 and all(1 <= i <= 40 for i in flat)          # every word its own id, inside the range
    assert any("collides" in str(w.message) for w in rec)
    # order-independent once the vocabulary was pre-scanned (records as the drivers hold them)
    recs = [{"original_prompt": f"[{p}]", "editing_prompt": p} for p in prompts]
    b, c = Small(stable_ids=True), Small(stable_ids=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert prescan_prompts(b, recs) == 30 and prescan_prompts(c, list(reversed(recs))) == 30
        assert [b.encode(p) for p in prompts] == [c.encode(p) for p in reversed(prompts)][::-1]
    assert prescan_prompts(object(), recs) == 0                                 # real tokenizers: untouched


def test_python_controller_protocol_of_the_base_class():
    """A subclass of hedit's AttentionControl that overrides the reference's forward() (ptp_classes.py:81-108) is a Python
    controller: the base class's __call__ hands forward() the conditional half only, writes its result back in place and
    counts layers / steps like the reference; the oracle's controller (pinned on the reference's in g4) counts the same
    call sequence the same way."""
    from hedit.p2p.ptp_classes import AttentionControl, AttentionStore, EmptyControl, runs_in_python
    from oracle import p2p as OP

    class Halve(AttentionControl):
        def __init__(self):
            super().__init__()
            self.calls, self.between = [], 0

        def forward(self, attn, is_cross, place_in_unet, save_attn):
            self.calls.append((tuple(attn.shape), is_cross, place_in_unet))
            return attn * 0.5 if is_cross else attn

        def between_steps(self):
            self.between += 1

    c = Halve()
    assert runs_in_python(c) and runs_in_python(lambda *a: None)
    assert not runs_in_python(EmptyControl()) and not runs_in_python(AttentionStore()) and not runs_in_python(None)
    c.num_att_layers = 3
    probs = hash_probs((8, 16, 77), 5)
    before = probs.clone()
    for i, (is_cross, place) in enumerate(((True, "down"), (False, "mid"), (True, "up"))):
        ret = c(probs, is_cross, place, True)
        assert ret is probs
        assert c.cur_att_layer == (i + 1) % 3
    assert c.cur_step == 1 and c.between == 1
    assert [s[0] for s in c.calls] == [(4, 16, 77)] * 3                      # the conditional half
    assert torch.equal(probs[:4], before[:4])                               # the unconditional half is never touched
    assert torch.allclose(probs[4:], before[4:] * 0.25)                      # two cross layers, each halved in place
    # save_attn=False: the edit applies, the counters stand still (ptp_classes.py:100-101)
    c(probs, True, "down", False)
    assert (c.cur_step, c.cur_att_layer) == (1, 0) and len(c.calls) == 4
    # the oracle's controller counts the same way on the same call sequence
    oc = OP.Controller("store")
    oc.num_att_layers = 3
    p2 = before.clone()
    for is_cross, place in ((True, "down"), (False, "mid"), (True, "up")):
        oc(p2, is_cross, place, True)
    assert (oc.cur_step, oc.cur_att_layer) == (1, 0)


@pytest.mark.parametrize("pi", range(len(PROMPT_PAIRS)))
def test_subclass_calling_super_forward_reproduces_reference_controller(golden_dir, pi):
    """The reference's extension idiom -- subclass AttentionStore / AttentionReplace / AttentionRefine / AttentionReweight,
    override forward() and call super().forward(...) (ptp_classes.py:135-150, 202-227) -- runs on the hook path, where the
    edit has to happen in Python.  Driven with the call sequence that produced g4 (the reference's own controllers), the
    edited target rows, the untouched rows, the counters and the stored map must match the reference's."""
    from hedit.p2p import ptp_classes as PC
    g = _npz(golden_dir, "g4_controller.npz")
    info = _json(golden_dir, "g4_controller.json")["pairs"][pi]
    tok = WordTokenizer(split_long_words_at=6)
    base = make(PROMPT_PAIRS[pi], 50, info["eq_val"], tok)
    seen = []

    def fwd(self, attn, is_cross, place_in_unet, save_attn):
        seen.append((is_cross, place_in_unet))
        return super(Sub, self).forward(attn, is_cross, place_in_unet, save_attn)
    Sub = type("Sub", (type(base),), {"forward": fwd})
    c = base
    c.__class__ = Sub                         # same tables, the subclass's forward()
    assert PC.runs_in_python(c) and not PC.runs_in_python(make(PROMPT_PAIRS[pi], 50, info["eq_val"], tok))
    c.num_att_layers = 4
    heads = info["heads"]
    for cur_step in (0, 16, 17, 19, 20, 49):
        c.cur_step, c.cur_att_layer = cur_step, 0
        c.step_store, c.attention_store = c.get_empty_store(), {}
        for li, (is_cross, place, n) in enumerate(info["layers"]):
            probs = hash_probs((4 * heads, n, 77 if is_cross else n), 100000 + pi * 1000 + cur_step * 10 + li)
            before = probs.clone()
            c(probs, is_cross, place, True)
            want = torch.from_numpy(g[f"p{pi}_s{cur_step}_l{li}_tar"])
            assert (probs[3 * heads:] - want).abs().max().item() <= 2e-6, (cur_step, li)
            assert torch.equal(probs[:3 * heads], before[:3 * heads])
        assert [c.cur_step, c.cur_att_layer] == info["after"][str(cur_step)]
        assert {k: len(v) for k, v in c.attention_store.items()} == info["store_counts"][str(cur_step)]
        assert (c.attention_store["down_cross"][0] - torch.from_numpy(g[f"p{pi}_s{cur_step}_store_down_cross0"])).abs().max() <= 2e-6
    # save_attn=False: the edit applies, nothing is stored, the counters stand still
    c.cur_step, c.cur_att_layer = 3, 0
    c.step_store, c.attention_store = c.get_empty_store(), {}
    probs = hash_probs((4 * heads, 16, 77), 900000 + pi)
    c(probs, True, "down", False)
    assert (probs[3 * heads:] - torch.from_numpy(g[f"p{pi}_nosave_tar"])).abs().max().item() <= 2e-6
    assert [c.cur_step, c.cur_att_layer, sum(len(v) for v in c.step_store.values())] == info["nosave_after"]
    assert len(seen) == 6 * 4 + 1


def test_store_subclass_accumulates_over_steps_like_the_reference():
    """AttentionStore.forward / between_steps on the hook path: maps of <= 32 x 32 tokens are summed over steps per layer
    (ptp_classes.py:135-150); the fused path never fills step_store, so between_steps is a no-op there."""
    from hedit.p2p.ptp_classes import AttentionStore, runs_in_python
    from oracle import p2p as OP

    class Keep(AttentionStore):
        def forward(self, attn, is_cross, place_in_unet, save_attn=True):
            return super().forward(attn, is_cross, place_in_unet, save_attn)

    c, oc = Keep(), OP.Controller("store")
    assert runs_in_python(c)
    c.num_att_layers = oc.num_att_layers = 3
    for step in range(3):
        for li, (is_cross, place, n) in enumerate(((True, "down", 16), (False, "mid", 40 * 40), (True, "up", 32))):
            probs = hash_probs((4, n, 77 if is_cross else 8), 7000 + 10 * step + li)
            p2 = probs.clone()
            c(probs, is_cross, place, True)
            oc(p2, is_cross, place, True)
    assert c.cur_step == oc.cur_step == 3
    assert {k: len(v) for k, v in c.attention_store.items()} == {k: len(v) for k, v in oc.attention_store.items()}
    assert len(c.attention_store["mid_self"]) == 0                     # 1600 tokens: above the 32 x 32 limit
    for k in c.attention_store:
        for a, b in zip(c.attention_store[k], oc.attention_store[k]):
            assert torch.equal(a, b)
    plain = AttentionStore()
    plain.between_steps()                                              # fused path: nothing in step_store, nothing happens
    assert plain.attention_store == {}
