"""Pin oracle/ against the golden vectors produced by running the reference's own modules
(tests/golden/make_golden.py).  CPU only."""
import json
import os
import types

import numpy as np
import pytest
import torch

from helpers.tiny import (PROMPT_PAIRS, TinyAttention, ddim_tables, hash_normal, hash_probs,
                          hash_uniform, make_tiny_model)
from oracle import loops as OL
from oracle import p2p as OP
from oracle import sched as OS

torch.set_num_threads(4)


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _json(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def close(a, b, tol=1e-5):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float32)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(1.0, ref), f"max abs err {err} (ref max {ref})"


# --------------------------------------------------------------------------- G1 / G2
@pytest.mark.parametrize("T", [10, 20, 50])
def test_scheduler_tables(golden_dir, T):
    g = _json(golden_dir, "g1_scheduler.json")[str(T)]
    sch = ddim_tables(T)
    assert [int(t) for t in sch.timesteps] == g["timesteps"]
    for row in g["rows"]:
        t, tt = row["t"], row["tt"]
        assert abs(float(OS.get_variance(sch, t)) - row["variance"]) <= 1e-7 + 1e-6 * abs(row["variance"])
        for eta in (0.0, 1.0):
            for ddim in (False, True):
                want = row[f"coeff_eta{int(eta)}_ddim{int(ddim)}"]
                got = float(OS.full_coeff(sch, t, tt, eta, ddim))
                assert abs(got - want) <= 1e-6, (t, tt, eta, ddim, got, want)
        # survey invariant 3: ddim-inversion coefficient is sqrt(1-abar_prev)
        assert abs(float(OS.full_coeff(sch, t, tt, 1.0, True)) -
                   float((1 - sch.alphas_cumprod[tt]) ** 0.5)) <= 1e-7


def test_reverse_step(golden_dir):
    g = _npz(golden_dir, "g2_reverse_step.npz")
    sch = ddim_tables(20)
    eps, x, z = (torch.from_numpy(g[k]) for k in ("eps", "x", "z"))
    for t in (951, 501, 1):
        for eta in (0.0, 1.0):
            for ddim in (False, True):
                prev, x0 = OS.reverse_step(sch, eps, t, x, eta=eta, z=z, ddim_inv=ddim, want_x0=True)
                close(prev, g[f"prev_t{t}_eta{int(eta)}_ddim{int(ddim)}"])
                close(x0, g[f"x0_t{t}_eta{int(eta)}_ddim{int(ddim)}"])
        close(OS.tweedie_x0(sch, eps, t, x), g[f"tweedie_t{t}"])


# --------------------------------------------------------------------------- G4 controller
def build_oracle_controller(model, pair, num_steps, xa=0.4, sa=0.35, eq_val=2.0):
    src, tar, blend, is_replace = pair
    bw = ((blend[0],), (blend[1],)) if blend else None
    eq = {"words": (blend[1],), "values": (eq_val,)} if blend else None
    return OP.make_controller([src, tar], is_replace, xa, sa, blend_word=bw, eq_params=eq,
                              num_steps=num_steps, tok=model.tokenizer)


@pytest.mark.parametrize("pi", range(len(PROMPT_PAIRS)))
def test_controller_tables_and_edits(golden_dir, pi):
    g = _npz(golden_dir, "g4_controller.npz")
    meta = _json(golden_dir, "g4_controller.json")
    info = meta["pairs"][pi]
    T = 50
    model = make_tiny_model(T)
    c = build_oracle_controller(model, PROMPT_PAIRS[pi], T, eq_val=info["eq_val"])
    assert model.tokenizer.encode(info["src"]) == info["src_ids"]
    assert np.array_equal(c.mapper.numpy(), g[f"p{pi}_mapper"])
    if f"p{pi}_alphas" in g:
        close(c.alphas, g[f"p{pi}_alphas"], 0)
    if f"p{pi}_equalizer" in g:
        close(c.eq, g[f"p{pi}_equalizer"], 0)
    else:
        assert c.eq is None
    close(c.cross_alpha, g[f"p{pi}_cross_replace_alpha"], 0)
    assert list(c.self_window) == info["num_self_replace"]
    if c.local_blend is not None:
        close(c.local_blend.alpha_layers, g[f"p{pi}_lb_alpha_layers"], 0)
        assert c.local_blend.start_blend == info["lb_start_blend"]
    for key, want in info["word_inds"].items():
        text, w = key.split("|")
        assert [int(v) for v in OP.word_inds(text, w, model.tokenizer)] == want

    heads = info["heads"]
    c.num_att_layers = 4
    for cur_step in (0, 16, 17, 19, 20, 49):
        c.cur_step, c.cur_att_layer = cur_step, 0
        c.step_store, c.attention_store = OP._empty_store(), {}
        for li, (is_cross, place, n) in enumerate(info["layers"]):
            k = 77 if is_cross else n
            probs = hash_probs((4 * heads, n, k), 100000 + pi * 1000 + cur_step * 10 + li)
            before = probs.clone()
            c(probs, is_cross, place, True)
            key = f"p{pi}_s{cur_step}_l{li}"
            close(probs[3 * heads:], g[key + "_tar"], 1e-6)
            # survey invariant 4: uncond half + source quarter untouched
            assert torch.equal(before[:3 * heads], probs[:3 * heads]) == info["rest_unchanged"][key]
        assert [c.cur_step, c.cur_att_layer] == info["after"][str(cur_step)]
        assert {k_: len(v) for k_, v in c.attention_store.items()} == info["store_counts"][str(cur_step)]
        close(c.attention_store["down_cross"][0], g[f"p{pi}_s{cur_step}_store_down_cross0"], 1e-6)
    c.cur_step, c.cur_att_layer = 3, 0
    c.step_store, c.attention_store = OP._empty_store(), {}
    probs = hash_probs((4 * heads, 16, 77), 900000 + pi)
    c(probs, True, "down", False)
    close(probs[3 * heads:], g[f"p{pi}_nosave_tar"], 1e-6)
    assert [c.cur_step, c.cur_att_layer, sum(len(v) for v in c.step_store.values())] == info["nosave_after"]


def test_big_self_attention_not_replaced(golden_dir):
    meta = _json(golden_dir, "g4_controller.json")
    model = make_tiny_model(50)
    c = build_oracle_controller(model, PROMPT_PAIRS[0], 50)
    c.num_att_layers = 1
    probs = hash_probs((4, 1089, 1089), 77)
    before = probs.clone()
    c(probs, False, "down", True)
    assert torch.equal(before, probs) == meta["big_self_unchanged"] is True
    assert sum(len(v) for v in c.attention_store.values()) == meta["big_self_stored"] == 0


# --------------------------------------------------------------------------- G5 LocalBlend
@pytest.mark.parametrize("pi", [0, 1, 3])
def test_local_blend(golden_dir, pi):
    g = _npz(golden_dir, "g5_local_blend.npz")
    T = 10
    model = make_tiny_model(T)
    c = build_oracle_controller(model, PROMPT_PAIRS[pi], T)
    heads = 2
    five = [hash_uniform((2 * heads, 256, 77), 3000 + pi * 10 + i) ** 6 for i in range(5)]
    big = torch.zeros(2 * heads, 1024, 77)
    store = {"down_cross": [big, big, five[0], five[1]], "up_cross": [five[2], five[3], five[4], big]}
    x = hash_normal((2, 4, 64, 64), 3500 + pi)
    for counter in (0, 2, 3):
        c.local_blend.counter = counter
        y = c.local_blend(x.clone(), store)
        assert torch.equal(y[0], x[0])
        close(y[1], g[f"p{pi}_y_counter{counter}"], 1e-6)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_local_blend_substruct_words(golden_dir, ci):
    """LocalBlend(substruct_words=...) (ptp_classes.py:28-38,64-68) against the reference's class (g16)."""
    from helpers.tiny import LOCAL_BLEND_SUB_CASES
    from oracle import p2p as OP
    g = _npz(golden_dir, "g16_local_blend_sub.npz")
    pi, words, sub, th = LOCAL_BLEND_SUB_CASES[ci]
    T = 10
    model = make_tiny_model(T)
    src, tar = PROMPT_PAIRS[pi][:2]
    lb = OP.LocalBlend([src, tar], T, words, model.tokenizer, th=th, sub_words=sub)
    heads = 2
    five = [hash_uniform((2 * heads, 256, 77), 3100 + ci * 10 + i) ** 6 for i in range(5)]
    big = torch.zeros(2 * heads, 1024, 77)
    store = {"down_cross": [big, big, five[0], five[1]], "up_cross": [five[2], five[3], five[4], big]}
    x = hash_normal((2, 4, 64, 64), 3600 + ci)
    lb.counter = 5
    y = lb(x.clone(), store)
    assert torch.equal(y[0], x[0])
    close(y[1], g[f"c{ci}_y"], 1e-6)
    assert int(g[f"c{ci}_cut_pixels"]) > 0


# --------------------------------------------------------------------------- G6 processor
def test_processor(golden_dir):
    g = _npz(golden_dir, "g6_processor.npz")
    T = 50
    model = make_tiny_model(T)
    c = build_oracle_controller(model, PROMPT_PAIRS[1], T)
    c.num_att_layers = 2
    gen = torch.Generator().manual_seed(400)
    a_self = TinyAttention(64, None, 8, gen)
    a_cross = TinyAttention(64, 32, 8, gen)
    for nm, mod in (("self", a_self), ("cross", a_cross)):
        mod.load_state_dict({k: torch.from_numpy(g[f"{nm}.{k}"]) for k in mod.state_dict()})
    hs, ctx = hash_normal((4, 64, 64), 401), hash_normal((4, 77, 32), 402)
    pd, pu = OP.P2PProcessor(c, "down"), OP.P2PProcessor(c, "up")
    with torch.no_grad():
        c.cur_step = 0
        close(pd(a_self, hs, None, use_controller=True, save_attn=True), g["out_self_ctrl"])
        close(pu(a_cross, hs, ctx, use_controller=True, save_attn=True), g["out_cross_ctrl"])
        close(pd(a_self, hs, None, use_controller=False), g["out_self_off"])
        close(pu(a_cross, hs, ctx, use_controller=False), g["out_cross_off"])
        c.cur_step = 30
        close(pd(a_self, hs, None, use_controller=True, save_attn=False), g["out_self_late"])
        close(pu(a_cross, hs, ctx, use_controller=True, save_attn=False), g["out_cross_late"])


# --------------------------------------------------------------------------- G3 loops
@pytest.fixture(scope="module")
def loops_golden(golden_dir):
    return _npz(golden_dir, "g3_loops.npz"), _json(golden_dir, "g3_loops.json")


@pytest.mark.parametrize("pi", [0, 2])
def test_ddpm_inversion(loops_golden, pi):
    g, meta = loops_golden
    T = meta["T"]
    model = make_tiny_model(T)
    w0 = torch.from_numpy(g["w0"])
    torch.manual_seed(4321 + pi)
    zs, wts, noise = OL.ddpm_inversion(model, w0, eta=1.0, prompt=PROMPT_PAIRS[pi][0], cfg_src=1.0, T=T)
    close(noise, g[f"inv{pi}_noise"], 0)
    close(wts, g[f"inv{pi}_wts"], 2e-5)
    close(zs, g[f"inv{pi}_zs"], 2e-4)


LOOP_FNS = {"h_Edit_p2p_implicit": OL.h_edit_p2p_implicit, "h_Edit_p2p_explicit": OL.h_edit_p2p_explicit,
            "h_Edit_R_implicit": OL.h_edit_r_implicit, "h_Edit_R_explicit": OL.h_edit_r_explicit}


@pytest.mark.parametrize("ci", range(8))
def test_loops(loops_golden, ci):
    g, meta = loops_golden
    case = meta["cases"][ci]
    T = meta["T"]
    pi = case["pair"]
    model = make_tiny_model(T)
    zs, wts = torch.from_numpy(g[f"inv{pi}_zs"]), torch.from_numpy(g[f"inv{pi}_wts"])
    after = T - case["skip"]
    if case["p2p"]:
        c = build_oracle_controller(model, PROMPT_PAIRS[pi], after, eq_val=case["eq_val"])
    else:
        c = OP.Controller("store")
    OP.register(model, c)
    assert c.num_att_layers == case["num_att_layers"]
    kw = dict(eta=1.0, prompts=[PROMPT_PAIRS[pi][0], PROMPT_PAIRS[pi][1]], cfg_scales=[1.0, 5.0, 7.5],
              zs=zs[:after], controller=c, after_skip_steps=after, is_ddim_inversion=case["ddim"])
    if "implicit" in case["fn"]:
        kw.update(weight_reconstruction=case["wrec"], optimization_steps=case["K"])
    edit, recon = LOOP_FNS[case["fn"]](model, xT=wts[after], **kw)
    name = case["name"]
    close(recon, g[f"{name}_recon"], 2e-4)
    close(edit, g[f"{name}_edit"], 2e-4)
    assert c.cur_step == case["cur_step"]
    if "lb_counter" in case:
        assert c.local_blend.counter == case["lb_counter"]
    if case["p2p"]:
        maps = c.attention_store["down_cross"][2:4] + c.attention_store["up_cross"][:3]
        got = np.stack([m.reshape(2, -1, 256, 77).sum(1)[:, :, :16].numpy() for m in maps])
        close(got, g[f"{name}_maps"], 2e-4)
    # survey invariant 1: with P2P the x^orig branch replays the inversion chain, so it returns
    # the inverted input (up to the toy network's error amplification)
    if case["p2p"] and not case["ddim"]:
        close(recon[0], g["w0"][0], 2e-2)


# --------------------------------------------------------------------------- G7 DDIM inversion / h-Edit-D
def test_ddim_inversion_and_h_edit_d(golden_dir):
    g = _npz(golden_dir, "g7_ddim.npz")
    meta = _json(golden_dir, "g7_ddim.json")
    T = meta["T"]
    w0 = torch.from_numpy(g["w0"])
    inv = {}
    for case in meta["cases"]:
        pi = case["pair"]
        if pi not in inv:
            model = make_tiny_model(T)
            model.scheduler = ddim_tables(T, steps_offset=0)
            lat, zs, lats = OL.ddim_inversion(model, w0, PROMPT_PAIRS[pi][0], case["cfg_src"])
            close(zs, g[f"inv{pi}_zs"], 2e-5)
            close(torch.stack([l[0] for l in lats]), g[f"inv{pi}_lats"], 2e-5)
            inv[pi] = (zs, lats)
        zs, lats = inv[pi]
        model = make_tiny_model(T)
        model.scheduler = ddim_tables(T, steps_offset=0)
        after = T - case["skip"]
        src, tar, blend, is_replace = PROMPT_PAIRS[pi]
        bw = ((blend[0],), (blend[1],)) if blend else None
        eq = {"words": (blend[1],), "values": (2.0,)} if blend else None
        c = OP.make_controller([src, tar], is_replace, 0.4, 0.6, blend_word=bw, eq_params=eq, num_steps=after,
                               tok=model.tokenizer)
        OP.register(model, c)
        kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[case["cfg_src"], 5.0, 7.5], zs=zs[:after], controller=c,
                  after_skip_steps=after, is_ddim_inversion=True)
        if "implicit" in case["fn"]:
            kw.update(weight_reconstruction=0.1, optimization_steps=case["K"])
        edit, recon = LOOP_FNS[case["fn"]](model, xT=lats[after], **kw)
        close(edit, g[case["name"] + "_edit"], 2e-4)
        close(recon, g[case["name"] + "_recon"], 2e-4)
        # the x^orig branch replays the deterministic inversion exactly
        close(recon, w0, 2e-3)
