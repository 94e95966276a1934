"""Host-language controllers through the executor's attention hook (hedit_unet_set_attn_hook): the reference's hook point
`self.controller(attention_probs, is_cross, self.place_in_unet, save_attn)` (text-guided/p2p/ptp_utils.py:98-106) with a
controller that is NOT one of hedit's.  The foreign controller used here is the oracle's restatement of the reference's
controller classes (oracle/p2p.py::Controller, a checker; ptp_classes.py:91-283) with its tables moved to the GPU -- so the
same edit runs once inside the fused kernels (hedit's own controller) and once as Python on materialised probabilities,
and the two must agree; against the CPU oracle the hooked path must hold the tolerance of the fused one."""
import ctypes as C
import math

import pytest
import torch

import helpers.gpu as G
from helpers.tiny import PROMPT_PAIRS
from hedit import _lib
from hedit.unet import TINY_CONFIG, AttnProcessor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return _lib.lib()


@pytest.fixture(scope="module")
def tiny():
    from helpers.models import make_pair
    return make_pair(TINY_CONFIG, 10)


def _inputs(B, cfg, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["in_channels"], cfg["sample_size"], cfg["sample_size"], generator=g)
    ctx = torch.randn(B, 77, cfg["cross_attention_dim"], generator=g)
    return x, ctx


def _to_gpu(oc):
    for name in ("cross_alpha", "mapper", "alphas", "eq"):
        v = getattr(oc, name, None)
        if isinstance(v, torch.Tensor):
            setattr(oc, name, v.cuda())
    if oc.local_blend is not None:
        oc.local_blend.alpha_layers = oc.local_blend.alpha_layers.cuda()
    return oc


def _plain(hip):
    hip.unet.set_attn_processor({k: AttnProcessor() for k in hip.unet.attn_processors})


@pytest.mark.parametrize("d,heads,N,M,kstride", [(40, 8, 256, 256, 256), (40, 8, 100, 77, 80), (160, 2, 64, 77, 80),
                                                   (32, 2, 1024, 1024, 1024), (80, 4, 72, 72, 72)])
def test_probabilities_and_apply_kernels(lib, d, heads, N, M, kstride):
    """softmax(q k^T) as fp32 [B*heads][N][M] in the reference's layout, then probs . v, against fp32 torch."""
    B, Cc = 2, d * heads
    g = torch.Generator().manual_seed(N + d)
    q = G.bf(torch.randn(B, N, Cc, generator=g) * d ** -0.5 * math.log2(math.e) * 1.5)
    k = G.bf(torch.randn(B, kstride, Cc, generator=g) * 1.5)
    v = G.bf(torch.randn(B, kstride, Cc, generator=g))
    vt = v.reshape(B * kstride, Cc).t().contiguous()
    probs = torch.empty(B * heads, N, M, dtype=torch.float32, device=G.dev())
    out = torch.zeros(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    _lib.check(lib.hedit_k_attn_probs(_lib.ptr(q), Cc, _lib.ptr(k), Cc, _lib.ptr(probs), B, N, M, kstride, heads, d, None))
    _lib.check(lib.hedit_k_attn_apply(_lib.ptr(probs), _lib.ptr(vt), B * kstride, _lib.ptr(out), Cc, B, N, M, kstride, heads, d, None))
    G.sync()
    qh = q.float().reshape(B, N, heads, d).transpose(1, 2)
    kh = k.float()[:, :M].reshape(B, M, heads, d).transpose(1, 2)
    vh = v.float()[:, :M].reshape(B, M, heads, d).transpose(1, 2)
    want_p = torch.softmax((qh @ kh.transpose(-1, -2)) * math.log(2.0), dim=-1)
    assert G.max_err(probs, want_p.reshape(B * heads, N, M)) < 2e-5
    assert float((probs.sum(-1) - 1).abs().max()) < 1e-5
    want = (want_p @ vh).transpose(1, 2).reshape(B, N, Cc)
    assert G.rel_err(out.float(), want) < 4e-3          # bf16 output rounding only: the probabilities stay fp32


def test_identity_controller_sees_every_layer_and_matches_the_fused_path(tiny):
    """A controller that only looks: called once per attention layer in execution order with the reference's shapes;
    the result is the fused kernels' up to their bf16 probabilities."""
    hip, om, _ = tiny
    calls = []

    def spy(attn, is_cross, place, save_attn):
        assert attn.dtype == torch.float32 and attn.is_cuda and save_attn is True
        assert float((attn.sum(-1) - 1).abs().max()) < 1e-4
        calls.append((tuple(attn.shape), is_cross, place))

    from hedit.p2p.ptp_utils import register_attention_control
    holder = type("H", (), {"__call__": staticmethod(spy), "num_att_layers": -1})()
    x, ctx = _inputs(4, TINY_CONFIG, 5)
    try:
        base = hip.unet(G.f32(x), 301, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"use_controller": False}).sample
        register_attention_control(hip, holder)
        got = hip.unet(G.f32(x), 301, encoder_hidden_states=G.f32(ctx)).sample
        G.sync()
    finally:
        _plain(hip)
    assert holder.num_att_layers == len(calls) == 22
    heads = TINY_CONFIG["attention_head_dim"]
    places = [c[2] for c in calls]
    assert places == ["down"] * 8 + ["mid"] * 2 + ["up"] * 12
    for i, (shape, is_cross, _) in enumerate(calls):
        assert is_cross == bool(i % 2) and shape[0] == 4 * heads
        assert shape[2] == (77 if is_cross else shape[1])
    assert calls[0][0][1] == 32 * 32 and calls[8][0][1] == 8 * 8
    with torch.no_grad():
        want = om.unet(x, torch.tensor(301), encoder_hidden_states=ctx, cross_attention_kwargs={"use_controller": False}).sample
    e_hook, e_fused = G.rel_err(got, want), G.rel_err(base, want)
    assert e_hook < 2.5e-2 and e_fused < 2.5e-2, (e_hook, e_fused)       # the tolerance of one eps evaluation (test_gpu_unet.py)
    assert G.rel_err(got, base) < 2.5e-2
    # use_controller=False leaves the hook alone: the fused path again, bit for bit
    again = hip.unet(G.f32(x), 301, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"use_controller": False}).sample
    assert torch.equal(again, base)


@pytest.mark.parametrize("pi,cur_step", [(0, 0), (2, 1), (3, 5)])
def test_reference_protocol_controller_matches_the_in_kernel_edit(tiny, pi, cur_step):
    """The same P2P pass three ways: hedit's controller (edit inside the kernels), the reference-protocol controller as
    Python through the hook, and the CPU oracle.  eps, the stored maps -- the self maps too, which only the hooked path
    keeps (ptp_classes.py:135-150) -- and the counters."""
    from oracle import p2p as OP
    from hedit.p2p import ptp_controller_utils as PCU
    from hedit.p2p.ptp_utils import register_attention_control
    hip, om, _ = tiny
    src, tar, blend, is_replace = PROMPT_PAIRS[pi]
    T = 10
    bw = ((blend[0],), (blend[1],)) if blend else None
    eq = {"words": (blend[1],), "values": (2.0,)} if blend else None
    hc = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=T,
                             tokenizer=hip.tokenizer, device=hip.device)
    fc = _to_gpu(OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=T, tok=om.tokenizer))
    oc = OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=T, tok=om.tokenizer)
    x, ctx = _inputs(4, TINY_CONFIG, 77 + pi)
    x[2], x[3] = x[0], x[1]
    try:
        register_attention_control(hip, hc)
        hc.cur_step = cur_step
        fused = hip.unet(G.f32(x), 401, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"save_attn": True}).sample
        register_attention_control(hip, fc)
        assert fc.num_att_layers == 22
        fc.cur_step = cur_step
        hooked = hip.unet(G.f32(x), 401, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"save_attn": True}).sample
        G.sync()
        OP.register(om, oc)
        oc.cur_step = cur_step
        with torch.no_grad():
            want = om.unet(x, torch.tensor(401), encoder_hidden_states=ctx, cross_attention_kwargs={"save_attn": True}).sample
    finally:
        _plain(hip)
        from oracle.sd_unet import PlainProcessor
        om.unet.set_attn_processor({k: PlainProcessor() for k in om.unet.attn_processors})
    # (the two GPU paths differ by what an identity controller already shows: bf16 against fp32 probabilities, ~1.5e-2 on
    #  this random network; both hold the tolerance of one eps evaluation against the oracle)
    assert G.rel_err(hooked, fused) < 2.5e-2
    assert G.rel_err(hooked, want) < 2.5e-2
    assert fc.cur_step == oc.cur_step == hc.cur_step == cur_step + 1 and fc.cur_att_layer == 0
    for key in ("down_cross", "mid_cross", "up_cross", "down_self", "mid_self", "up_self"):
        assert len(fc.attention_store[key]) == len(oc.attention_store[key])
        for a, b in zip(fc.attention_store[key], oc.attention_store[key]):
            assert a.shape == b.shape and G.rel_err(a, b) < 2e-2
    for key in ("down_cross", "mid_cross", "up_cross"):
        for a, b in zip(fc.attention_store[key], hc.attention_store[key]):
            assert G.rel_err(a, b) < 2e-2


@pytest.mark.parametrize("pi,cur_step", [(0, 0), (2, 1), (3, 5)])
def test_fused_path_keeps_the_self_maps_on_request(tiny, pi, cur_step):
    """store_self_maps = True: the fused path materialises the <= 32 x 32 SELF maps of the conditional rows beside the flash
    kernel and accumulates them like the reference's AttentionStore (ptp_classes.py:135-150): same keys, counts, shapes and
    values as the CPU oracle's store over two passes (the second adds to the first: between_steps), post-edit inside the
    self-replace window (the target row's map is the source's); eps is untouched, bit for bit."""
    from oracle import p2p as OP
    from hedit.p2p import ptp_controller_utils as PCU
    from hedit.p2p.ptp_utils import register_attention_control
    hip, om, _ = tiny
    src, tar, blend, is_replace = PROMPT_PAIRS[pi]
    T = 10
    bw = ((blend[0],), (blend[1],)) if blend else None
    eq = {"words": (blend[1],), "values": (2.0,)} if blend else None
    mk = lambda: PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=T,    # noqa: E731
                                     tokenizer=hip.tokenizer, device=hip.device)
    hc, plain = mk(), mk()
    hc.store_self_maps = True
    oc = OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=T, tok=om.tokenizer)
    x, ctx = _inputs(4, TINY_CONFIG, 177 + pi)
    x[2], x[3] = x[0], x[1]
    try:
        register_attention_control(hip, plain)
        plain.cur_step = cur_step
        base = [hip.unet(G.f32(x), t, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"save_attn": True}).sample.clone()
                for t in (401, 381)]
        register_attention_control(hip, hc)
        hc.cur_step = cur_step
        got = [hip.unet(G.f32(x), t, encoder_hidden_states=G.f32(ctx), cross_attention_kwargs={"save_attn": True}).sample.clone()
               for t in (401, 381)]
        G.sync()
        OP.register(om, oc)
        oc.cur_step = cur_step
        with torch.no_grad():
            for t in (401, 381):
                om.unet(x, torch.tensor(t), encoder_hidden_states=ctx, cross_attention_kwargs={"save_attn": True})
    finally:
        _plain(hip)
        from oracle.sd_unet import PlainProcessor
        om.unet.set_attn_processor({k: PlainProcessor() for k in om.unet.attn_processors})
    assert all(torch.equal(a, b) for a, b in zip(got, base))               # storing never changes the evaluation
    assert hc.cur_step == oc.cur_step == cur_step + 2
    assert all(len(plain.attention_store[k]) == 0 for k in ("down_self", "mid_self", "up_self"))
    n_self = 0
    for key in ("down_cross", "mid_cross", "up_cross", "down_self", "mid_self", "up_self"):
        assert len(hc.attention_store[key]) == len(oc.attention_store[key]), key
        for a, b in zip(hc.attention_store[key], oc.attention_store[key]):
            assert tuple(a.shape) == tuple(b.shape) and G.rel_err(a, b) < 2e-2, (key, G.rel_err(a, b))
            n_self += key.endswith("self")
    assert n_self == 11                                                     # every transformer block of the tiny UNet is <= 32 x 32


def test_sampling_loop_with_a_foreign_controller(tiny):
    """h_Edit_p2p_implicit end to end with the reference-protocol controller hooked (LocalBlend through step_callback on
    the maps the hook stored) against the same loop with hedit's in-kernel controller."""
    from oracle import loops as OL
    from oracle import p2p as OP
    from hedit.inversion.p2p_h_edit import h_Edit_p2p_implicit
    from hedit.p2p import ptp_controller_utils as PCU
    from hedit.p2p.ptp_utils import register_attention_control
    from helpers.models import make_pair
    T = 4
    hip, om, _ = make_pair(TINY_CONFIG, T, out_scale=0.3)
    src, tar, blend, is_replace = PROMPT_PAIRS[0]
    torch.manual_seed(0)
    w0 = torch.randn(1, 4, 32, 32) * 0.8
    zs, wts, _ = OL.ddpm_inversion(om, w0, eta=1.0, prompt=src, cfg_src=1.0, T=T)
    bw = ((blend[0],), (blend[1],))
    eq = {"words": (blend[1],), "values": (2.0,)}
    hc = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, equilizer_params=eq, num_steps=T,
                             tokenizer=hip.tokenizer, device=hip.device)
    fc = _to_gpu(OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=T, tok=om.tokenizer))
    kw = dict(eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], weight_reconstruction=0.1, optimization_steps=2,
              after_skip_steps=T, is_ddim_inversion=False)
    try:
        register_attention_control(hip, hc)
        e_n, r_n = h_Edit_p2p_implicit(hip, xT=wts[T].cuda(), zs=zs.cuda(), controller=hc, **kw)
        register_attention_control(hip, fc)
        e_f, r_f = h_Edit_p2p_implicit(hip, xT=wts[T].cuda(), zs=zs.cuda(), controller=fc, **kw)
        G.sync()
    finally:
        _plain(hip)
    oc = OP.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=bw, eq_params=eq, num_steps=T, tok=om.tokenizer)
    OP.register(om, oc)
    try:
        with torch.no_grad():
            e_o, r_o = OL.h_edit_p2p_implicit(om, xT=wts[T], zs=zs, controller=oc, **kw)
    finally:
        from oracle.sd_unet import PlainProcessor
        om.unet.set_attn_processor({k: PlainProcessor() for k in om.unet.attn_processors})
    assert fc.cur_step == hc.cur_step == oc.cur_step
    assert G.rel_err(r_f, r_n) < 2e-2 and G.rel_err(r_f, r_o) < 3e-2
    # the random-weight sampler chain amplifies every bf16 rounding (smoke(): 1e-1 against the oracle for this loop)
    err_f, err_n = G.rel_err(e_f, e_o), G.rel_err(e_n, e_o)
    assert err_f < 1e-1 and err_n < 1e-1, (err_f, err_n)
    assert G.rel_err(e_f, e_n) < 1.2e-1


def test_subclass_with_a_python_forward_runs_through_the_hook(tiny):
    """The reference's extension point: subclass AttentionControl, override forward() (ptp_classes.py:81-108).  hedit's base
    class does the reference's slicing (conditional half only) and counting; a subclass WITHOUT forward() stays on the fused path."""
    from hedit.p2p.ptp_classes import AttentionControl, EmptyControl, runs_in_python
    from hedit.p2p.ptp_utils import register_attention_control
    hip, om, _ = tiny

    class SwapCross(AttentionControl):
        """target rows take the source rows' cross maps (what AttentionReplace does with an identity mapper and alpha = 1)"""

        def __init__(self):
            super().__init__()
            self.seen = []
            self.steps_seen = 0

        def forward(self, attn, is_cross, place_in_unet, save_attn):
            self.seen.append((tuple(attn.shape), is_cross, place_in_unet))
            if is_cross:
                h = attn.shape[0] // 2
                attn = attn.clone()
                attn[h:] = attn[:h]
            return attn

        def between_steps(self):
            self.steps_seen += 1

    c = SwapCross()
    assert runs_in_python(c) and not runs_in_python(EmptyControl()) and not runs_in_python(None)
    x, ctx = _inputs(4, TINY_CONFIG, 31)
    x[2], x[3] = x[0], x[1]
    heads = TINY_CONFIG["attention_head_dim"]
    try:
        register_attention_control(hip, c)
        got = hip.unet(G.f32(x), 201, encoder_hidden_states=G.f32(ctx)).sample
        G.sync()
    finally:
        _plain(hip)
    assert len(c.seen) == 22 and all(s[0][0] == 2 * heads for s in c.seen)            # the conditional half: [src, tar] x heads
    assert (c.cur_step, c.cur_att_layer, c.steps_seen) == (1, 0, 1)
    # the same edit on the oracle's reference-shaped processor path
    from oracle import p2p as OP

    class OSwap:
        num_att_layers = -1

        def __call__(self, probs, is_cross, place, save_attn):
            if is_cross:
                h = probs.shape[0]
                q = h // 4
                probs[3 * q:] = probs[2 * q:3 * q]
            return probs

    oc = OSwap()
    OP.register(om, oc)
    try:
        with torch.no_grad():
            want = om.unet(x, torch.tensor(201), encoder_hidden_states=ctx).sample
    finally:
        from oracle.sd_unet import PlainProcessor
        om.unet.set_attn_processor({k: PlainProcessor() for k in om.unet.attn_processors})
    assert G.rel_err(got, want) < 2.5e-2


def test_a_failing_controller_surfaces_and_the_hook_is_released(tiny, lib):
    hip, _, _ = tiny
    from hedit.p2p.ptp_utils import register_attention_control

    class Boom:
        num_att_layers = -1
        n = 0

        def __call__(self, attn, is_cross, place, save_attn):
            self.n += 1
            if self.n == 3:
                raise ValueError("controller says no")

    x, ctx = _inputs(2, TINY_CONFIG, 9)
    b = Boom()
    try:
        register_attention_control(hip, b)
        with pytest.raises(ValueError, match="controller says no"):
            hip.unet(G.f32(x), 5, encoder_hidden_states=G.f32(ctx))
        assert b.n == 3
        # not callable -> refused at registration
        with pytest.raises(TypeError):
            hip.unet.set_attn_processor({k: type("P", (), {"controller": 3})() for k in hip.unet.attn_processors})
    finally:
        _plain(hip)
    base = hip.unet(G.f32(x), 5, encoder_hidden_states=G.f32(ctx)).sample
    G.sync()
    assert torch.isfinite(base).all()
    # C level: a hook next to a plan with in-kernel edits is refused
    fn = hip.unet._HOOK_T(lambda *a: 0)
    _lib.check(lib.hedit_unet_set_attn_hook(hip.unet._h, C.cast(fn, C.c_void_p), None))
    try:
        plan = _lib.P2PPlan()
        plan.mode = 1
        plan.n_single = 2
        out = torch.empty(2, 4, 32, 32, device=G.dev())
        ws = hip.unet._workspace(2, 32, 32)
        rc = lib.hedit_unet_forward(hip.unet._h, _lib.ptr(G.f32(x)), C.c_float(5.0), _lib.ptr(G.f32(ctx)), 2, 32, 32, C.byref(plan),
                                    _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)
        assert rc != 0 and b"hook" in lib.hedit_last_error()
    finally:
        _lib.check(lib.hedit_unet_set_attn_hook(hip.unet._h, None, None))
