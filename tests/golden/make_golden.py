#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  It puts the reference's
``text-guided`` sub-project on sys.path, installs ``sys.modules`` stubs for the third-party
imports that are missing here and are import-time only on the hot path (diffusers, cv2, nltk),
imports the reference modules UNMODIFIED and drives them with the seeded toy objects of
``tests/helpers/tiny.py``.  Outputs are small .npz/.json files = data (inputs + expected
outputs); no reference source text is written anywhere.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz, *.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/text-guided"
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; golden vectors can only be (re)generated "
                         "in the build container")
    _stub("diffusers")
    _stub("diffusers.utils")
    _stub("diffusers.utils.torch_utils", randn_tensor=None)
    _stub("diffusers.models")
    _stub("diffusers.models.attention_processor", Attention=object)
    _stub("cv2")
    nltk = _stub("nltk", download=lambda *a, **k: None)
    tok = _stub("nltk.tokenize", word_tokenize=lambda s: s.split())
    nltk.tokenize = tok
    sys.path.insert(0, REF)
    import inversion.inversion_utils as iu
    import inversion.ddpm_inversion as di
    import inversion.ddim_inversion as dd
    import inversion.p2p_h_edit as he
    import p2p.ptp_utils as pu
    import p2p.ptp_classes as pc
    import p2p.seq_aligner as sa
    import p2p.ptp_controller_utils as pcu
    # tqdm progress bars off
    he.tqdm = lambda x, *a, **k: x
    di.tqdm = lambda x, *a, **k: x
    dd.tqdm = lambda x, *a, **k: x
    return types.SimpleNamespace(iu=iu, di=di, dd=dd, he=he, pu=pu, pc=pc, sa=sa, pcu=pcu)


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


# --------------------------------------------------------------------------- G1 / G2
def gen_scheduler(ref, out):
    from helpers.tiny import ddim_tables
    rec = {}
    for T in (10, 20, 50):
        sch = ddim_tables(T)
        model = types.SimpleNamespace(scheduler=sch, device=torch.device("cpu"))
        ts = [int(t) for t in sch.timesteps]
        rows = []
        for i, t in enumerate(ts):
            tt = ts[i + 1] if i + 1 < len(ts) else 0
            row = {"t": t, "tt": tt, "variance": float(ref.iu.get_variance(model, t)),
                   "alpha_bar_t": float(sch.alphas_cumprod[t])}
            for eta in (0.0, 1.0):
                for ddim in (False, True):
                    c = ref.iu.compute_full_coeff(model, t, tt, eta, ddim)
                    row[f"coeff_eta{int(eta)}_ddim{int(ddim)}"] = float(c)
            rows.append(row)
        rec[str(T)] = {"timesteps": ts, "final_alpha_cumprod": float(sch.final_alpha_cumprod),
                       "rows": rows}
    with open(os.path.join(out, "g1_scheduler.json"), "w") as f:
        json.dump(rec, f, indent=0)

    # G2 reverse_step
    sch = ddim_tables(20)
    model = types.SimpleNamespace(scheduler=sch, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(101)
    eps = torch.randn(2, 4, 8, 8, generator=g)
    x = torch.randn(2, 4, 8, 8, generator=g)
    z = torch.randn(4, 8, 8, generator=g)
    d = {"eps": npy(eps), "x": npy(x), "z": npy(z)}
    for t in (951, 501, 1):
        for eta in (0.0, 1.0):
            for ddim in (False, True):
                prev, x0 = ref.iu.reverse_step(model, eps, t, x, eta=eta, variance_noise=z,
                                               return_pred_x0=True, is_ddim_inversion=ddim)
                d[f"prev_t{t}_eta{int(eta)}_ddim{int(ddim)}"] = npy(prev)
                d[f"x0_t{t}_eta{int(eta)}_ddim{int(ddim)}"] = npy(x0)
        d[f"tweedie_t{t}"] = npy(ref.iu.reverse_step_pred_x0(model, eps, t, x))
    np.savez_compressed(os.path.join(out, "g2_reverse_step.npz"), **d)


# --------------------------------------------------------------------------- G4 / G5 tables+controller
def build_ref_controller(ref, model, pair, num_steps, xa=0.4, sa=0.35, eq_val=2.0):
    # (sa = 0.6 is the reference's self-replace fraction for h-Edit-D, main_p2p.py:70)
    src, tar, blend, is_replace = pair
    prompts = [src, tar]
    blend_word = ((blend[0],), (blend[1],)) if blend else None
    eq = {"words": (blend[1],), "values": (eq_val,)} if blend else None
    return ref.pcu.make_controller(prompts=prompts, is_replace_controller=is_replace,
                                   cross_replace_steps=xa, self_replace_steps=sa,
                                   blend_word=blend_word, equilizer_params=eq,
                                   num_steps=num_steps, tokenizer=model.tokenizer,
                                   device=model.device)


def gen_controller(ref, out):
    from helpers.tiny import make_tiny_model, PROMPT_PAIRS, hash_probs
    d = {}
    meta = {"pairs": []}
    T = 50
    for pi, pair in enumerate(PROMPT_PAIRS):
        model = make_tiny_model(T)
        ctrl = build_ref_controller(ref, model, pair, T, eq_val=2.0 if pi % 2 == 0 else 1.25)
        src, tar, blend, is_replace = pair
        info = {"src": src, "tar": tar, "blend": list(blend) if blend else None,
                "is_replace": is_replace, "eq_val": 2.0 if pi % 2 == 0 else 1.25,
                "class": type(ctrl).__name__,
                "src_ids": model.tokenizer.encode(src), "tar_ids": model.tokenizer.encode(tar)}
        base = ctrl.prev_controller if hasattr(ctrl, "prev_controller") and ctrl.prev_controller is not None else ctrl
        info["base_class"] = type(base).__name__
        d[f"p{pi}_mapper"] = npy(base.mapper)
        if hasattr(base, "alphas"):
            d[f"p{pi}_alphas"] = npy(base.alphas)
        if hasattr(ctrl, "equalizer"):
            d[f"p{pi}_equalizer"] = npy(ctrl.equalizer)
        d[f"p{pi}_cross_replace_alpha"] = npy(ctrl.cross_replace_alpha)
        info["num_self_replace"] = list(ctrl.num_self_replace)
        if ctrl.local_blend is not None:
            d[f"p{pi}_lb_alpha_layers"] = npy(ctrl.local_blend.alpha_layers)
            info["lb_start_blend"] = ctrl.local_blend.start_blend
        # word index tables
        wi = {}
        for text in (src, tar):
            for w in set(text.split(" ")):
                wi[f"{text}|{w}"] = [int(v) for v in ref.pu.get_word_inds(text, w, model.tokenizer)]
        info["word_inds"] = wi

        # controller numerics: one "UNet pass" = 4 attention layers; inputs are regenerated
        # from hash_probs(seed) by the tests, outputs: only the target-conditional quarter is
        # stored, plus a flag that every other row came back bit-identical
        ctrl.num_att_layers = 4
        heads = 1
        layers = [(True, "down", 16), (False, "down", 16), (True, "mid", 32), (False, "up", 32)]
        info["layers"] = [[c, pl, n] for c, pl, n in layers]
        info["heads"] = heads
        for cur_step in (0, 16, 17, 19, 20, 49):
            ctrl.cur_step = cur_step
            ctrl.cur_att_layer = 0
            ctrl.step_store = ctrl.get_empty_store()
            ctrl.attention_store = {}
            for li, (is_cross, place, n) in enumerate(layers):
                k = 77 if is_cross else n
                seed = 100000 + pi * 1000 + cur_step * 10 + li
                probs = hash_probs((4 * heads, n, k), seed)
                before = probs.clone()
                ctrl(probs, is_cross, place, True)
                key = f"p{pi}_s{cur_step}_l{li}"
                d[key + "_tar"] = npy(probs[3 * heads:]).astype(np.float32)
                info.setdefault("rest_unchanged", {})[key] = bool(
                    torch.equal(before[:3 * heads], probs[:3 * heads]))
            info.setdefault("after", {})[str(cur_step)] = [ctrl.cur_step, ctrl.cur_att_layer]
            # what the store holds after the pass (sum over everything as a cheap fingerprint
            # + the first stored cross map in full)
            st = ctrl.attention_store
            info.setdefault("store_counts", {})[str(cur_step)] = {k_: len(v) for k_, v in st.items()}
            d[f"p{pi}_s{cur_step}_store_down_cross0"] = npy(st["down_cross"][0]).astype(np.float32)
        # save_attn=False pass: edits applied, counters untouched, nothing stored
        ctrl.cur_step, ctrl.cur_att_layer = 3, 0
        ctrl.step_store = ctrl.get_empty_store()
        ctrl.attention_store = {}
        probs = hash_probs((4 * heads, 16, 77), 900000 + pi)
        ctrl(probs, True, "down", False)
        d[f"p{pi}_nosave_tar"] = npy(probs[3 * heads:])
        info["nosave_after"] = [ctrl.cur_step, ctrl.cur_att_layer,
                                sum(len(v) for v in ctrl.step_store.values())]
        meta["pairs"].append(info)

    # self-attention above the 32x32 threshold is never replaced (ptp_classes.py:194-200)
    model = make_tiny_model(T)
    ctrl = build_ref_controller(ref, model, PROMPT_PAIRS[0], T)
    ctrl.num_att_layers = 1
    ctrl.cur_step = 0
    probs = hash_probs((4, 1089, 1089), 77)
    before = probs.clone()
    ctrl(probs, False, "down", True)
    meta["big_self_unchanged"] = bool(torch.equal(before, probs))
    meta["big_self_stored"] = sum(len(v) for v in ctrl.attention_store.values()) if ctrl.attention_store else 0

    np.savez_compressed(os.path.join(out, "g4_controller.npz"), **d)
    with open(os.path.join(out, "g4_controller.json"), "w") as f:
        json.dump(meta, f, indent=0)


def gen_local_blend(ref, out):
    from helpers.tiny import make_tiny_model, PROMPT_PAIRS, hash_uniform, hash_normal
    d = {}
    T = 10
    for pi in (0, 1, 3):
        model = make_tiny_model(T)
        ctrl = build_ref_controller(ref, model, PROMPT_PAIRS[pi], T)
        lb = ctrl.local_blend
        heads = 2
        # the five 16x16 cross maps (peaky so the 0.3 threshold gives a non-trivial mask);
        # tests regenerate them with the same seeds
        five = [hash_uniform((2 * heads, 256, 77), 3000 + pi * 10 + i) ** 6 for i in range(5)]
        big = torch.zeros(2 * heads, 1024, 77)
        store = {"down_cross": [big, big, five[0], five[1]],
                 "up_cross": [five[2], five[3], five[4], big]}
        x = hash_normal((2, 4, 64, 64), 3500 + pi)
        for counter in (0, 2, 3):
            lb.counter = counter
            y = lb(x.clone(), store)
            d[f"p{pi}_y_counter{counter}"] = npy(y[1]).astype(np.float32)
            assert torch.equal(y[0], x[0])
    np.savez_compressed(os.path.join(out, "g5_local_blend.npz"), **d)


def gen_local_blend_sub(ref, out):
    """LocalBlend with substruct_words (ptp_classes.py:28-38,64-68), the reference's class called directly (its
    make_controller never passes them)."""
    from helpers.tiny import make_tiny_model, PROMPT_PAIRS, LOCAL_BLEND_SUB_CASES, hash_uniform, hash_normal
    d = {}
    T = 10
    for ci, (pi, words, sub, th) in enumerate(LOCAL_BLEND_SUB_CASES):
        model = make_tiny_model(T)
        src, tar = PROMPT_PAIRS[pi][:2]
        lb = ref.pc.LocalBlend([src, tar], T, words, substruct_words=sub, th=th, tokenizer=model.tokenizer, device=model.device)
        heads = 2
        five = [hash_uniform((2 * heads, 256, 77), 3100 + ci * 10 + i) ** 6 for i in range(5)]
        big = torch.zeros(2 * heads, 1024, 77)
        store = {"down_cross": [big, big, five[0], five[1]], "up_cross": [five[2], five[3], five[4], big]}
        x = hash_normal((2, 4, 64, 64), 3600 + ci)
        lb.counter = 5
        y = lb(x.clone(), store)
        assert torch.equal(y[0], x[0])
        d[f"c{ci}_y"] = npy(y[1]).astype(np.float32)
        lb0 = ref.pc.LocalBlend([src, tar], T, words, th=th, tokenizer=model.tokenizer, device=model.device)
        lb0.counter = 5
        y0 = lb0(x.clone(), store)
        d[f"c{ci}_cut_pixels"] = np.asarray(int((y0[1] != y[1]).any(0).sum()))      # how much the substruct mask removed
    np.savez_compressed(os.path.join(out, "g16_local_blend_sub.npz"), **d)


# --------------------------------------------------------------------------- G6 processor
def gen_processor(ref, out):
    from helpers.tiny import TinyAttention, make_tiny_model, PROMPT_PAIRS, hash_normal
    T = 50
    model = make_tiny_model(T)
    ctrl = build_ref_controller(ref, model, PROMPT_PAIRS[1], T)
    ctrl.num_att_layers = 2
    g = torch.Generator().manual_seed(400)
    a_self = TinyAttention(64, None, 8, g)
    a_cross = TinyAttention(64, 32, 8, g)
    hs = hash_normal((4, 64, 64), 401)
    ctx = hash_normal((4, 77, 32), 402)
    d = {}
    for nm, mod in (("self", a_self), ("cross", a_cross)):
        for k, v in mod.state_dict().items():
            d[f"{nm}.{k}"] = npy(v)
    proc_d = ref.pu.P2PCrossAttnProcessor(ctrl, "down")
    proc_u = ref.pu.P2PCrossAttnProcessor(ctrl, "up")
    with torch.no_grad():
        ctrl.cur_step = 0
        d["out_self_ctrl"] = npy(proc_d(a_self, hs, None, use_controller=True, save_attn=True))
        d["out_cross_ctrl"] = npy(proc_u(a_cross, hs, ctx, use_controller=True, save_attn=True))
        d["out_self_off"] = npy(proc_d(a_self, hs, None, use_controller=False))
        d["out_cross_off"] = npy(proc_u(a_cross, hs, ctx, use_controller=False))
        ctrl.cur_step = 30
        d["out_self_late"] = npy(proc_d(a_self, hs, None, use_controller=True, save_attn=False))
        d["out_cross_late"] = npy(proc_u(a_cross, hs, ctx, use_controller=True, save_attn=False))
    np.savez_compressed(os.path.join(out, "g6_processor.npz"), **d)


# --------------------------------------------------------------------------- G3 loops
def gen_loops(ref, out):
    from helpers.tiny import make_tiny_model, PROMPT_PAIRS
    T = 10
    d = {}
    meta = {"T": T, "cases": []}
    cfg = [1.0, 5.0, 7.5]

    def fresh():
        return make_tiny_model(T)

    model = fresh()
    torch.manual_seed(1234)
    w0 = torch.randn(1, 4, 16, 16) * 0.8
    d["w0"] = npy(w0)
    inv = {}
    for pi in (0, 2):
        model = fresh()
        torch.manual_seed(4321 + pi)
        _, zs, wts, noise = ref.di.inversion_forward_process_ddpm(
            model, w0, etas=1.0, prog_bar=False, prompt=PROMPT_PAIRS[pi][0], cfg_scale_src=1.0,
            num_inference_steps=T)
        inv[pi] = (zs, wts)
        d[f"inv{pi}_zs"] = npy(zs)
        d[f"inv{pi}_wts"] = npy(wts)
        d[f"inv{pi}_noise"] = npy(noise)

    def run(name, fn_name, pi, skip, K, ddim, p2p, wrec, eq_val=2.0):
        model = fresh()
        zs, wts = inv[pi]
        after = T - skip
        if p2p:
            ctrl = build_ref_controller(ref, model, PROMPT_PAIRS[pi], after, eq_val=eq_val)
        else:
            ctrl = ref.pc.AttentionStore()
        ref.pu.register_attention_control(model, ctrl)
        fn = getattr(ref.he, fn_name)
        kw = dict(eta=1.0, prompts=[PROMPT_PAIRS[pi][0], PROMPT_PAIRS[pi][1]], cfg_scales=cfg,
                  prog_bar=False, zs=zs[:after], controller=ctrl, after_skip_steps=after,
                  is_ddim_inversion=ddim)
        if "implicit" in fn_name:
            kw.update(weight_reconstruction=wrec, optimization_steps=K)
        edit, recon = fn(model, xT=wts[after], **kw)
        d[f"{name}_edit"] = npy(edit)
        d[f"{name}_recon"] = npy(recon)
        case = {"name": name, "fn": fn_name, "pair": pi, "skip": skip, "K": K, "ddim": ddim,
                "p2p": p2p, "wrec": wrec, "eq_val": eq_val, "cur_step": ctrl.cur_step,
                "num_att_layers": ctrl.num_att_layers}
        if p2p and ctrl.local_blend is not None:
            case["lb_counter"] = ctrl.local_blend.counter
        if p2p:
            # the five 16x16 cross maps LocalBlend reads, accumulated over the run
            maps = ctrl.attention_store["down_cross"][2:4] + ctrl.attention_store["up_cross"][:3]
            # (5 maps) x (src,tar) x 256 pixels x first 16 tokens, summed over heads
            d[f"{name}_maps"] = np.stack(
                [npy(m).reshape(2, -1, 256, 77).sum(1)[:, :, :16] for m in maps]).astype(np.float32)
        meta["cases"].append(case)

    run("p2p_imp_k1", "h_Edit_p2p_implicit", 0, 0, 1, False, True, 0.1)
    run("p2p_imp_k3_skip2", "h_Edit_p2p_implicit", 2, 2, 3, False, True, 0.1, eq_val=1.25)
    run("p2p_imp_k2_ddim", "h_Edit_p2p_implicit", 0, 0, 2, True, True, 0.075)
    run("p2p_exp", "h_Edit_p2p_explicit", 0, 0, 1, False, True, 0.1)
    run("p2p_exp_skip3_ddim", "h_Edit_p2p_explicit", 2, 3, 1, True, True, 0.1)
    run("r_imp_k2", "h_Edit_R_implicit", 0, 0, 2, False, False, 0.1)
    run("r_imp_k1_skip3", "h_Edit_R_implicit", 2, 3, 1, False, False, 0.1)
    run("r_exp", "h_Edit_R_explicit", 0, 0, 1, False, False, 0.1)
    np.savez_compressed(os.path.join(out, "g3_loops.npz"), **d)
    with open(os.path.join(out, "g3_loops.json"), "w") as f:
        json.dump(meta, f, indent=0)


def gen_ddim(ref, out):
    """h-Edit-D: DDIM inversion (eta = 0 scheduler of main_p2p.py:139-141: steps_offset 0) and the
    P2P loops run on its outputs with is_ddim_inversion=True, eta=1."""
    from helpers.tiny import make_tiny_model, PROMPT_PAIRS, ddim_tables
    T = 10
    d = {}
    meta = {"T": T, "cases": []}

    def fresh():
        m = make_tiny_model(T)
        m.scheduler = ddim_tables(T, steps_offset=0)
        return m

    torch.manual_seed(99)
    w0 = torch.randn(1, 4, 16, 16) * 0.8
    d["w0"] = npy(w0)
    for pi, cfg_src in ((0, 1.0), (2, 3.0)):
        model = fresh()
        lat, zs, lats = ref.dd.ddim_inversion(model, w0, PROMPT_PAIRS[pi][0], cfg_src)
        d[f"inv{pi}_zs"] = npy(zs)
        d[f"inv{pi}_lats"] = np.stack([npy(l)[0] for l in lats])
        for fn_name, K, skip in (("h_Edit_p2p_implicit", 1, 0), ("h_Edit_p2p_implicit", 2, 3), ("h_Edit_p2p_explicit", 1, 2)):
            model = fresh()
            after = T - skip
            ctrl = build_ref_controller(ref, model, PROMPT_PAIRS[pi], after, sa=0.6)
            ref.pu.register_attention_control(model, ctrl)
            kw = dict(eta=1.0, prompts=[PROMPT_PAIRS[pi][0], PROMPT_PAIRS[pi][1]], cfg_scales=[cfg_src, 5.0, 7.5],
                      prog_bar=False, zs=zs[:after], controller=ctrl, after_skip_steps=after, is_ddim_inversion=True)
            if "implicit" in fn_name:
                kw.update(weight_reconstruction=0.1, optimization_steps=K)
            edit, recon = getattr(ref.he, fn_name)(model, xT=lats[after], **kw)
            name = f"p{pi}_{fn_name}_k{K}_s{skip}"
            d[name + "_edit"] = npy(edit)
            d[name + "_recon"] = npy(recon)
            meta["cases"].append({"name": name, "fn": fn_name, "pair": pi, "cfg_src": cfg_src, "K": K, "skip": skip})
    np.savez_compressed(os.path.join(out, "g7_ddim.npz"), **d)
    with open(os.path.join(out, "g7_ddim.json"), "w") as f:
        json.dump(meta, f, indent=0)


# --------------------------------------------------------------------------- G8
# --------------------------------------------------------------------------- G13: MasaCtrl
def gen_masactrl(ref, out):
    """The reference's MutualSelfAttentionControl + regiter_attention_editor_diffusers + h_Edit_masactrl_implicit
    (text-guided/masactrl/, inversion/masactrl_h_edit.py), UNMODIFIED, on the toy UNet whose module hierarchy the
    registration walks (helpers.tiny.TinyMasaUNet).  The reference imports its own package under a second name
    (`masa_ctrl`): aliased here; torchvision.utils.save_image (unused on this path) is stubbed."""
    from helpers.tiny import make_tiny_masa_model, PROMPT_PAIRS
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    import masactrl.masactrl_utils as mu
    pkg = _stub("masa_ctrl")
    pkg.masactrl_utils = mu
    sys.modules["masa_ctrl.masactrl_utils"] = mu
    import masactrl.masactrl as mm
    import inversion.masactrl_h_edit as mh
    mh.tqdm = lambda x, *a, **k: x
    T = 10
    d = {}
    torch.manual_seed(1234)
    w0 = torch.randn(1, 4, 16, 16) * 0.8
    d["w0"] = npy(w0)
    meta = []
    for name, pi, skip, K, ddim, step, layer in (("masa_k1", 0, 0, 1, False, 2, 3), ("masa_k2_skip2", 2, 2, 2, False, 1, 5),
                                                 ("masa_ddim", 0, 0, 1, True, 0, 0), ("masa_off", 0, 0, 1, False, 99, 3)):
        model = make_tiny_masa_model(T)
        torch.manual_seed(4321 + pi)
        if ddim:
            _, zs, wts = ref.dd.ddim_inversion(model, w0, PROMPT_PAIRS[pi][0], 1.0)
        else:
            _, zs, wts, _ = ref.di.inversion_forward_process_ddpm(model, w0, etas=1.0, prog_bar=False, prompt=PROMPT_PAIRS[pi][0],
                                                                  cfg_scale_src=1.0, num_inference_steps=T)
        d[f"{name}_zs"], d[f"{name}_wts"] = npy(zs), npy(wts)
        model = make_tiny_masa_model(T)
        editor = mm.MutualSelfAttentionControl(step, layer)
        mu.regiter_attention_editor_diffusers(model, editor)
        after = T - skip
        edit, recon = mh.h_Edit_masactrl_implicit(model, xT=wts[after], eta=1.0, prompts=[PROMPT_PAIRS[pi][0], PROMPT_PAIRS[pi][1]],
                                                  cfg_scales=[1.0, 5.0, 7.5], prog_bar=False, zs=zs[:after], optimization_steps=K,
                                                  after_skip_steps=after, is_ddim_inversion=ddim)
        d[f"{name}_edit"], d[f"{name}_recon"] = npy(edit), npy(recon)
        meta.append({"name": name, "pair": pi, "skip": skip, "K": K, "ddim": ddim, "start_step": step, "start_layer": layer,
                     "cur_step": editor.cur_step, "num_att_layers": editor.num_att_layers})
    np.savez_compressed(os.path.join(out, "g13_masactrl.npz"), **d)
    with open(os.path.join(out, "g13_masactrl.json"), "w") as f:
        json.dump(meta, f, indent=0)


# --------------------------------------------------------------------------- G14: Plug-and-Play
def gen_pnp(ref, out):
    """The reference's Plug-and-Play hooks (plug_n_play/pnp_utils.py) and h_Edit_PnP_implicit (inversion/pnp_h_edit.py),
    UNMODIFIED, patched into the ORACLE's SD UNet (oracle/sd_unet.py exposes the diffusers attribute surface the hooks
    read) at the four-level toy configuration; weights / text encoder are the seeded ones the tests rebuild."""
    from helpers.tiny import make_oracle_sd_model, PROMPT_PAIRS, TINY4_CONFIG
    import plug_n_play.pnp_utils as pu
    import inversion.pnp_h_edit as ph
    ph.tqdm = lambda x, *a, **k: x
    T = 4
    d = {}
    torch.manual_seed(77)
    w0 = torch.randn(1, 4, 64, 64) * 0.8
    d["w0"] = npy(w0)
    meta = []
    for name, K, f_t, attn_t in (("pnp_k1", 1, 0.5, 0.5), ("pnp_k2_attn_only", 2, -1.0, 0.75)):
        model, _ = make_oracle_sd_model(TINY4_CONFIG, T)
        torch.manual_seed(300)
        _, zs, wts, _ = ref.di.inversion_forward_process_ddpm(model, w0, etas=1.0, prog_bar=False, prompt=PROMPT_PAIRS[0][0],
                                                              cfg_scale_src=1.0, num_inference_steps=T)
        d[f"{name}_zs"], d[f"{name}_wts"] = npy(zs), npy(wts)
        model, _ = make_oracle_sd_model(TINY4_CONFIG, T)
        n_f, n_a = int(T * f_t), int(T * attn_t)
        qk = model.scheduler.timesteps[:n_a] if n_a >= 0 else []
        conv = model.scheduler.timesteps[:n_f] if n_f >= 0 else []
        pu.register_attention_control_efficient(model, qk)
        pu.register_conv_control_efficient(model, conv)
        edit, recon = ph.h_Edit_PnP_implicit(model, xT=wts[T], eta=1.0, prompts=[PROMPT_PAIRS[0][0], PROMPT_PAIRS[0][1]],
                                             cfg_scales=[1.0, 5.0, 7.5], prog_bar=False, zs=zs[:T], optimization_steps=K,
                                             after_skip_steps=T, is_ddim_inversion=False)
        d[f"{name}_edit"], d[f"{name}_recon"] = npy(edit), npy(recon)
        meta.append({"name": name, "K": K, "qk": [int(v) for v in qk], "conv": [int(v) for v in conv]})
    np.savez_compressed(os.path.join(out, "g14_pnp.npz"), **d)
    with open(os.path.join(out, "g14_pnp.json"), "w") as f:
        json.dump(meta, f, indent=0)


def synthetic_rgb(h, w, seed):
    """smooth-ish deterministic uint8 image from integer arithmetic only (regenerated by the test)"""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    ch = [((x * (3 + c) + y * (5 - c) + seed * 17) % 256 + ((x * y + c * 31) // 7) % 64) % 256 for c in range(3)]
    return np.stack(ch, -1).astype(np.uint8)


def gen_load512(ref, out):
    """reference load_512 (p2p/ptp_classes.py:351-373) on synthetic arrays; stores every 4th pixel
    of the result as the integer k with x = k / 127.5 - 1, plus the sum of all k."""
    cases = [(300, 500, 0, 0, 0, 0), (640, 480, 0, 0, 0, 0), (512, 512, 0, 0, 0, 0), (400, 600, 10, 20, 5, 7),
             (200, 200, 300, 0, 0, 0)]
    rec = {}
    for i, (h, w, l, r, t, b) in enumerate(cases):
        img = synthetic_rgb(h, w, i)
        x = ref.pc.load_512(img, l, r, t, b, torch.device("cpu"))
        k = torch.round((x + 1) * 127.5).to(torch.int64)
        assert torch.allclose(k.float() / 127.5 - 1, x, atol=1e-6)
        rec[f"case{i}"] = np.array([h, w, l, r, t, b], dtype=np.int64)
        rec[f"sub{i}"] = npy(k[0, :, ::4, ::4]).astype(np.uint8)
        rec[f"sum{i}"] = np.array([int(k.sum())], dtype=np.int64)
    np.savez_compressed(os.path.join(out, "g8_load512.npz"), **rec)


# --------------------------------------------------------------------------- G9: text + style loop
REF_STYLE = "/root/reference/text-guided-n-style"


def gen_style_child(out):
    """Runs in its own interpreter (the n-style sub-project reuses the package names ``inversion`` /
    ``p2p`` of text-guided): imports text-guided-n-style/inversion/h_edit.py UNMODIFIED and drives
    h_Edit_p2p_implicit (text + style editing) with the toy UNet / VAE / style encoder."""
    global REF
    REF = REF_STYLE
    _stub("diffusers")
    _stub("diffusers.utils")
    _stub("diffusers.utils.torch_utils", randn_tensor=None)
    _stub("diffusers.models")
    _stub("diffusers.models.attention_processor", Attention=object)
    _stub("cv2")
    nltk = _stub("nltk", download=lambda *a, **k: None)
    nltk.tokenize = _stub("nltk.tokenize", word_tokenize=lambda s: s.split())
    sys.path.insert(0, REF_STYLE)
    import warnings
    warnings.filterwarnings("ignore")      # autocast("cuda") without a GPU: disabled with a warning
    import inversion.h_edit as hs
    import inversion.ddpm_inversion as di
    import p2p.ptp_utils as pu
    import p2p.ptp_classes as pc
    import p2p.ptp_controller_utils as pcu
    hs.tqdm = lambda x, *a, **k: x
    di.tqdm = lambda x, *a, **k: x
    ref = types.SimpleNamespace(pcu=pcu, pc=pc, pu=pu)
    from helpers.tiny import make_tiny_model, PROMPT_PAIRS, TinyVae, TinyStyleEncoder
    T = 10
    d = {}
    meta = {"T": T, "cases": []}
    cfg = [1.0, 5.0, 7.5]
    torch.manual_seed(1234)                 # the g3 start sample: a well-conditioned trajectory
    w0 = torch.randn(1, 4, 16, 16) * 0.8
    d["w0"] = npy(w0)

    def run(name, pi, skip, K, weight, with_encoder=True, blend=True):
        model = make_tiny_model(T)
        model.vae = TinyVae()
        torch.manual_seed(4321 + pi)
        _, zs, wts, _ = di.inversion_forward_process_ddpm(model, w0, etas=1.0, prog_bar=False,
                                                          prompt=PROMPT_PAIRS[pi][0], cfg_scale_src=1.0,
                                                          num_inference_steps=T)
        d[f"{name}_zs"] = npy(zs)
        d[f"{name}_wts"] = npy(wts)
        model = make_tiny_model(T)          # fresh objects for the edit, as in gen_loops
        model.vae = TinyVae()
        after = T - skip
        pair = PROMPT_PAIRS[pi] if blend else PROMPT_PAIRS[pi][:2] + (None, PROMPT_PAIRS[pi][3])
        ctrl = build_ref_controller(ref, model, pair, after)
        pu.register_attention_control(model, ctrl)
        enc = TinyStyleEncoder() if with_encoder else None
        edit, recon = hs.h_Edit_p2p_implicit(model, enc, xT=wts[after], eta=1.0,
                                             prompts=[pair[0], pair[1]], cfg_scales=cfg, prog_bar=False,
                                             zs=zs[:after], controller=ctrl, weight_edit_clip=weight,
                                             optimization_steps=K, after_skip_steps=after, is_ddim_inversion=False)
        d[f"{name}_edit"] = npy(edit)
        d[f"{name}_recon"] = npy(recon)
        meta["cases"].append({"name": name, "pair": pi, "skip": skip, "K": K, "weight": weight,
                              "with_encoder": with_encoder, "blend": blend, "cur_step": ctrl.cur_step})

    run("style_k1", 0, 0, 1, 0.5, blend=False)           # main_edit.py passes blend_word=None
    run("style_k2_skip2", 2, 2, 2, 0.55)
    run("style_blend", 2, 0, 1, 0.8)
    run("style_noenc", 0, 0, 1, 0.5, with_encoder=False)  # `if image_encoder:` false -> text editing only
    np.savez_compressed(os.path.join(out, "g9_style.npz"), **d)
    with open(os.path.join(out, "g9_style.json"), "w") as f:
        json.dump(meta, f, indent=0)
    gen_clip(out)


def name_seed(name):
    import zlib
    return zlib.crc32(name.encode()) % 100003


def gen_clip(out):
    """G10: the reference's CLIPEncoder.get_gram_matrix_residual (clip_guidance/base_clip.py:55-66) and its
    CLIP ViT (clip_guidance/clip/model.py, UNMODIFIED) at toy width: 4 blocks of width 64, patch 32 at
    the 224 x 224 the encoder always resizes to.  Third-party imports missing here are stubbed
    (torchvision.transforms: Normalize / ToTensor / Compose with their documented arithmetic; ftfy), the
    download is replaced by a locally built CLIP with hash-seeded weights, and Tensor.cuda() is the
    identity (no GPU in the build container).  Runs inside the n-style child interpreter."""
    from PIL import Image
    from helpers.tiny import hash_normal

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(-1, 1, 1)
            self.std = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean.to(x)) / self.std.to(x)

    class ToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.asarray(pic, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Normalize=Normalize, ToTensor=ToTensor, Compose=Compose,
                          Resize=object, CenterCrop=object, InterpolationMode=types.SimpleNamespace(BICUBIC=3))
    _stub("ftfy", fix_text=lambda s: s)
    sys.path.insert(0, REF_STYLE)
    import clip_guidance.base_clip as bc
    from clip_guidance.clip import model as cm
    clip = cm.CLIP(embed_dim=32, image_resolution=224, vision_layers=4, vision_width=64, vision_patch_size=32,
                   context_length=8, vocab_size=16, transformer_width=64, transformer_heads=1, transformer_layers=1)
    with torch.no_grad():
        for name, p in clip.named_parameters():
            if not name.startswith("visual."):
                continue
            v = hash_normal(tuple(p.shape), name_seed(name))
            if name.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight")):
                v = 1.0 + 0.1 * v
            elif p.dim() == 1:
                v = 0.1 * v
            else:
                v = v * (float(p[0].numel()) ** -0.5 if p.dim() > 1 else 1.0)
            p.copy_(v)
    clip.eval()
    bc.load_clip_to_cpu = lambda: clip
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref_rgb = synthetic_rgb(40, 56, 3)
    tmp = os.path.join(out, "_style_ref_tmp.png")
    Image.fromarray(ref_rgb).save(tmp)
    enc = bc.CLIPEncoder(need_ref=True, ref_path=tmp)
    os.remove(tmp)
    d = {"ref_rgb": ref_rgb, "ref_tensor_sub": npy(enc.ref[0, :, ::8, ::8])}
    for i, hw in enumerate(((64, 64), (96, 80))):
        im = (hash_normal((1, 3) + hw, 900 + i) * 0.6).requires_grad_(True)
        res = enc.get_gram_matrix_residual(im)
        loss = torch.linalg.norm(res)
        (g,) = torch.autograd.grad(loss, im)
        x = torch.nn.functional.interpolate(im.detach(), size=(224, 224), mode="bicubic")
        _, feats = clip.encode_image_with_features(enc.preprocess(x))
        d[f"residual{i}"] = npy(res)
        d[f"loss{i}"] = np.array([loss.item()], dtype=np.float64)
        d[f"grad{i}"] = npy(g)
        d[f"feat{i}"] = npy(feats[2][:, 0, :])
    np.savez_compressed(os.path.join(out, "g10_clip.npz"), **d)


# --------------------------------------------------------------------------- G11: face-swapping path
REF_FACE = "/root/reference/face-swapping"
FACE_TINY = dict(type="simple", in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16],
                 dropout=0.0, image_size=32, resamp_with_conv=True, num_diffusion_timesteps=1000)


def face_state_dict(shapes):
    """hash-seeded weights by parameter name (regenerated identically by the tests)"""
    from helpers.tiny import hash_normal
    sd = {}
    for name, shape in shapes.items():
        v = hash_normal(tuple(shape), name_seed(name))
        if "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * v
        elif len(shape) == 1:
            v = 0.05 * v
        else:
            v = v * float(np.prod(shape[1:])) ** -0.5
        sd[name] = v
    return sd


def gen_face_child(out):
    """Runs in its own interpreter with face-swapping/ on sys.path: the reference's pixel UNet
    (diffusion/diffusion.py::Model), SDE inversion and h_Edit_R, all UNMODIFIED, at toy size."""
    sys.path.insert(0, REF_FACE)
    import warnings
    warnings.filterwarnings("ignore")
    from diffusion.diffusion import Model
    from diffusion.diffusion_utils import get_beta_schedule
    import inversion.h_edit_R as he
    import inversion.sde_inversion as si
    he.tqdm = lambda x, *a, **k: x
    si.tqdm = lambda x, *a, **k: x
    from helpers.tiny import hash_normal, TinyIdLoss, TinyLpips
    model = Model(dict(FACE_TINY)).eval()
    sd = face_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    d = {}
    x = hash_normal((2, 3, 32, 32), 321) * 0.8
    for t in (1, 501, 991):
        with torch.no_grad():
            d[f"unet_t{t}"] = npy(model(x, torch.ones(2) * t))
    betas = torch.from_numpy(get_beta_schedule(beta_schedule="linear", beta_start=0.0001, beta_end=0.02,
                                               num_diffusion_timesteps=1000)).float()
    T = 10
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    x0 = hash_normal((1, 3, 32, 32), 654) * 0.6
    _, zs, xts, _ = si.inversion_forward_process_sde(model, x0, betas, seq, etas=1.0, num_inference_steps=T, device="cpu")
    d["zs"], d["xts"] = npy(zs), npy(xts)
    idl, lp = TinyIdLoss(), TinyLpips()
    mask = (hash_normal((1, 1, 32, 32), 77) > 0).float()
    for name, skip, K, w, use_id, use_lp, m in (("face_k2", 0, 2, 4.0, True, True, None),
                                                ("face_k1_skip3_mask", 3, 1, 6.0, True, True, mask),
                                                ("face_idonly", 0, 1, 4.0, True, False, None),
                                                ("face_lponly", 2, 2, 4.0, False, True, None)):
        after = T - skip
        out_x = he.h_Edit_R(model, lp if use_lp else None, idl if use_id else None, xts[after].clone(), betas, seq, eta=1.0,
                            zs=zs[:after], weight_edit_face=w, optimization_steps=K, after_skip_steps=after,
                            num_inference_steps=T, soft_face_mask=m)
        d[name] = npy(out_x)
    d["mask"] = npy(mask)
    np.savez_compressed(os.path.join(out, "g11_face.npz"), **d)
    gen_idloss(out)


def irse_state_dict(shapes):
    """hash-seeded IR-SE50 weights / BatchNorm statistics by state_dict name (regenerated by the tests)"""
    from helpers.tiny import hash_normal
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros(shape, dtype=torch.long)
            continue
        v = hash_normal(tuple(shape) if len(shape) else (1,), name_seed(name)).reshape(shape)
        if name.endswith("running_var"):
            v = 1.0 + 0.2 * v.abs()
        elif len(shape) > 1:
            v = v * float(np.prod(shape[1:])) ** -0.5
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * v          # BatchNorm / PReLU scales
        else:
            v = 0.05 * v
        sd[name] = v
    return sd


def gen_idloss(out):
    """G12: the reference's IDLoss (face-swapping/arcface/arcface_model.py:11-67) with its IR-SE50 backbone,
    UNMODIFIED; torchvision.transforms.ToTensor and lpips are stubbed, torch.load returns hash-seeded weights
    instead of the checkpoint file, Tensor.cuda() is the identity."""
    from PIL import Image
    from helpers.tiny import hash_normal

    class ToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.asarray(pic, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)

    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", ToTensor=ToTensor)
    _stub("lpips", LPIPS=object)
    from arcface.facial_recognition.model_irse import Backbone
    probe = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se")
    sd = irse_state_dict({k: tuple(v.shape) for k, v in probe.state_dict().items()})
    real_load = torch.load
    torch.load = lambda *a, **k: sd
    torch.Tensor.cuda = lambda self, *a, **k: self
    import arcface.arcface_model as am
    ref_rgb = synthetic_rgb(70, 90, 9)
    tmp = os.path.join(out, "_face_ref_tmp.png")
    Image.fromarray(ref_rgb).save(tmp)
    idl = am.IDLoss(ref_path=tmp)
    os.remove(tmp)
    torch.load = real_load
    d = {"ref_rgb": ref_rgb, "ref_tensor_sub": npy(idl.ref[0, :, ::16, ::16])}
    for i, (b, hw) in enumerate(((1, 256), (2, 128))):
        x = (hash_normal((b, 3, hw, hw), 40 + i) * 0.4).requires_grad_(True)
        loss = idl.get_cosine_loss(x)
        (g,) = torch.autograd.grad(loss, x)
        d[f"feat{i}"] = npy(idl.extract_feats(x.detach()))
        d[f"sim{i}"] = npy(idl.get_cosine_sim(x.detach()))
        d[f"loss{i}"] = np.array([loss.item()], dtype=np.float64)
        d[f"grad_sub{i}"] = npy(g[:, :, ::4, ::4])
    np.savez_compressed(os.path.join(out, "g12_idloss.npz"), **d)


def gen_face(out):
    import subprocess
    subprocess.run([sys.executable, os.path.abspath(__file__), "--face-child"], check=True)


def gen_style(out):
    import subprocess
    subprocess.run([sys.executable, os.path.abspath(__file__), "--style-child"], check=True)


def main():
    torch.set_num_threads(4)
    torch.set_grad_enabled(True)
    if "--style-child" in sys.argv:
        if not os.path.isdir(REF_STYLE):
            raise SystemExit("reference tree not present")
        return gen_style_child(HERE)
    if "--face-child" in sys.argv:
        if not os.path.isdir(REF_FACE):
            raise SystemExit("reference tree not present")
        return gen_face_child(HERE)
    if "--only-style" in sys.argv:
        return gen_style(HERE)
    if "--only-face" in sys.argv:
        return gen_face(HERE)
    if "--only-pnp" in sys.argv:
        return gen_pnp(import_reference(), HERE)
    if "--only-local-blend-sub" in sys.argv:
        return gen_local_blend_sub(import_reference(), HERE)
    if "--only-masactrl" in sys.argv:
        return gen_masactrl(import_reference(), HERE)
    ref = import_reference()
    out = HERE
    gen_scheduler(ref, out)
    gen_controller(ref, out)
    gen_local_blend(ref, out)
    gen_local_blend_sub(ref, out)
    gen_processor(ref, out)
    gen_loops(ref, out)
    gen_ddim(ref, out)
    gen_load512(ref, out)
    gen_masactrl(ref, out)
    gen_pnp(ref, out)
    gen_style(out)
    gen_face(out)
    for f in sorted(os.listdir(out)):
        if f.endswith((".npz", ".json")):
            print(f"{f:32s} {os.path.getsize(os.path.join(out, f)) / 1024:9.1f} KiB")


if __name__ == "__main__":
    main()
