"""UNet2DConditionModel facade over the native HIP executor (csrc/unet.hip).

Exposes the surface the reference uses of diffusers' UNet:
``unet(sample, timestep, encoder_hidden_states=, cross_attention_kwargs=).sample`` /
``["sample"]`` (text-guided/inversion/p2p_h_edit.py:613, ddpm_inversion.py:130),
``in_channels`` / ``sample_size`` (ddpm_inversion.py:29-31), ``attn_processors`` /
``set_attn_processor`` (p2p/ptp_utils.py:280,294), ``zero_grad`` (main_p2p.py:277), and the
diffusers state_dict key names for weights.

The attention processors are markers: the Prompt-to-Prompt edit itself runs inside the HIP
attention kernels, driven by a per-call plan compiled from the registered controller
(hedit/p2p/ptp_classes.py).  A controller that is NOT one of hedit's (any object callable as the
reference's ``controller(attention_probs, is_cross, place_in_unet, save_attn)``,
p2p/ptp_utils.py:98-106, p2p/ptp_classes.py:91-108) is served by the hook path of the executor
(``hedit_unet_set_attn_hook``): every attention layer materialises its probabilities, the controller
sees and may rewrite them in place, the executor multiplies what comes back with V.  Slow (fp32
probabilities through HBM), same protocol.
"""
import ctypes as C
import hashlib
import math

import torch

from . import _lib

SD15_CONFIG = dict(in_channels=4, out_channels=4, sample_size=64,
                   block_out_channels=(320, 640, 1280, 1280),
                   down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                     "CrossAttnDownBlock2D", "DownBlock2D"),
                   up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                                   "CrossAttnUpBlock2D"),
                   layers_per_block=2, cross_attention_dim=768, attention_head_dim=8,
                   norm_num_groups=32)

# three levels at 32x32 latents, keeps SD's stored-cross-map inventory (tests / smoke)
TINY_CONFIG = dict(in_channels=4, out_channels=4, sample_size=32,
                   block_out_channels=(64, 128, 128),
                   down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                   up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                   layers_per_block=2, cross_attention_dim=64, attention_head_dim=2,
                   norm_num_groups=32)


class UNetOutput(dict):
    @property
    def sample(self):
        return self["sample"]


class AttnProcessor:
    """Plain attention (no controller).  Marker only."""
    controller = None


def random_state_dict(param_shapes, seed=0):
    """Synthetic weights (no checkpoints exist offline): W ~ N(0, 1/fan_in),
    small random biases / norm affine terms.  Each tensor is seeded by (seed, name), so the
    result does not depend on iteration order."""
    sd = {}
    for name, shape in param_shapes.items():
        hsh = hashlib.sha256(f"{seed}:{name}".encode()).digest()
        g = torch.Generator().manual_seed(int.from_bytes(hsh[:7], "little"))
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif "norm" in name and name.endswith("weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "norm" in name:
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            sd[name] = 0.02 * torch.randn(shape, generator=g)
    return sd


class UNet2DConditionModel:
    def __init__(self, config=None, device="cuda:0"):
        cfg = dict(SD15_CONFIG)
        cfg.update(config or {})
        self.config = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hedit.UNet2DConditionModel runs on the GPU only (HIP kernels)")
        self.in_channels = cfg["in_channels"]
        self.sample_size = cfg["sample_size"]
        self._lib = _lib.lib()
        c = _lib.UnetCfg()
        c.in_channels, c.out_channels, c.sample_size = cfg["in_channels"], cfg["out_channels"], cfg["sample_size"]
        ch = list(cfg["block_out_channels"])
        c.n_levels = len(ch)
        for i, v in enumerate(ch):
            c.block_out_channels[i] = v
            c.down_has_attn[i] = int(cfg["down_block_types"][i].startswith("CrossAttn"))
            c.up_has_attn[i] = int(cfg["up_block_types"][i].startswith("CrossAttn"))
        c.layers_per_block = cfg["layers_per_block"]
        c.cross_attention_dim = cfg["cross_attention_dim"]
        c.heads = cfg["attention_head_dim"]
        c.norm_num_groups = cfg["norm_num_groups"]
        self._cfg_struct = c
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_unet_create(C.byref(c), C.byref(h)))
        self._h = h
        self.param_shapes = {}
        self.bf16_exact = set()       # parameters the executor keeps as unscaled bf16 copies (hedit.dist sends those as bf16)
        nd, dims = C.c_int(), (C.c_int * 4)()
        for i in range(self._lib.hedit_unet_num_params(self._h)):
            name = self._lib.hedit_unet_param_name(self._h, i).decode()
            _lib.check(self._lib.hedit_unet_param_shape(self._h, i, C.byref(nd), dims))
            self.param_shapes[name] = tuple(dims[k] for k in range(nd.value))
            if self._lib.hedit_unet_param_bf16_exact(self._h, i):
                self.bf16_exact.add(name)
        self._procs = {n: AttnProcessor() for n in self._processor_names()}
        self._ws = None
        self.heads = cfg["attention_head_dim"]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.hedit_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.param_shapes if k not in sd]
        extra = [k for k in sd if k not in self.param_shapes]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]} ({len(missing)}), "
                           f"unexpected {extra[:5]} ({len(extra)})")
        st = _lib.cur_stream()
        with torch.cuda.device(self.device):
            for name, shape in self.param_shapes.items():
                if name not in sd:
                    continue
                w = sd[name]
                if tuple(w.shape) != shape:
                    raise ValueError(f"{name}: expected shape {shape}, got {tuple(w.shape)}")
                w = w.detach().to(device=self.device, dtype=torch.float32).contiguous()
                _lib.check(self._lib.hedit_unet_load(self._h, name.encode(), _lib.ptr(w), w.numel(), st))
                # the pack kernels are stream-ordered; keep `w` alive until they ran
                torch.cuda.current_stream().synchronize()
        return self

    def init_random(self, seed=0):
        sd = random_state_dict(self.param_shapes, seed)
        self.load_state_dict(sd)
        return sd

    def zero_grad(self, *a, **k):
        return None

    # ---------------------------------------------------------------- processor registry
    def _processor_names(self):
        cfg = self.config
        names = []
        L = cfg["layers_per_block"]
        for i, t in enumerate(cfg["down_block_types"]):
            if t.startswith("CrossAttn"):
                for j in range(L):
                    for a in ("attn1", "attn2"):
                        names.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
        for a in ("attn1", "attn2"):
            names.append(f"mid_block.attentions.0.transformer_blocks.0.{a}.processor")
        for i, t in enumerate(cfg["up_block_types"]):
            if t.startswith("CrossAttn"):
                for j in range(L + 1):
                    for a in ("attn1", "attn2"):
                        names.append(f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
        return names

    @property
    def attn_processors(self):
        return dict(self._procs)

    def set_attn_processor(self, procs):
        if not isinstance(procs, dict):
            procs = {k: procs for k in self._procs}
        for k, p in procs.items():
            if k not in self._procs:
                raise KeyError(f"unknown attention processor name {k}")
            if not hasattr(p, "controller"):
                raise TypeError(
                    f"{type(p).__name__}: a processor is identified by its .controller (None = plain attention, a hedit "
                    "controller = in-kernel edit, any other callable = hooked through hedit_unet_set_attn_hook); processor "
                    "objects that re-implement the attention body itself cannot be run -- the body is the HIP kernels")
            c = p.controller
            if c is not None and not hasattr(c, "_plan") and not callable(c):      # (hedit's own are callable too)
                raise TypeError(f"{type(c).__name__}: a foreign controller must be callable as "
                                "controller(attention_probs, is_cross, place_in_unet, save_attn)")
            self._procs[k] = p

    def _controller(self):
        ctrls = {id(p.controller): p.controller for p in self._procs.values() if p.controller is not None}
        if not ctrls:
            return None
        if len(ctrls) != 1 or any(p.controller is None for p in self._procs.values()):
            raise RuntimeError("all attention layers must share one controller "
                               "(use hedit.p2p.ptp_utils.register_attention_control)")
        return next(iter(ctrls.values()))

    # ---------------------------------------------------------------- geometry helpers
    def store_layers(self, height, width):
        """[(tokens, place)] of the cross-attention layers whose maps are stored (<= 32*32 tokens),
        in call order; place in {'down','mid','up'}."""
        n = self._lib.hedit_unet_num_store_layers(self._h, height, width)
        out = []
        tok, pl = C.c_int(), C.c_int()
        for i in range(n):
            _lib.check(self._lib.hedit_unet_store_layer_info(self._h, height, width, i, C.byref(tok), C.byref(pl)))
            out.append((tok.value, ("down", "mid", "up")[pl.value]))
        return out

    def _workspace(self, B, H, W):
        need = self._lib.hedit_unet_workspace_bytes(self._h, B, H, W)
        if need == 0:
            raise _lib.HipError("workspace planning failed: " + self._lib.hedit_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---------------------------------------------------------------- sampled launch timing
    PROF_KINDS = ("conv3x3_gemm", "linear_gemm", "self_attn", "cross_attn", "norm", "other")

    def prof_enable(self, on, max_records=8192):
        _lib.check(self._lib.hedit_prof_enable(self._h, int(on), max_records))

    def prof_reset(self):
        _lib.check(self._lib.hedit_prof_reset(self._h))

    def prof_collect(self):
        """{kind: (total_ms, total_flops, launches)}; synchronises the device first."""
        torch.cuda.synchronize(self.device)
        out = {}
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        for i, k in enumerate(self.PROF_KINDS):
            _lib.check(self._lib.hedit_prof_collect(self._h, i, C.byref(ms), C.byref(fl), C.byref(n)))
            out[k] = (ms.value, fl.value, n.value)
        return out

    def prof_collect_bytes(self):
        """{kind: algorithmic HBM bytes of the sampled launches} (operands read once, results written once)."""
        out = {}
        b = C.c_double()
        for i, k in enumerate(self.PROF_KINDS):
            _lib.check(self._lib.hedit_prof_collect_bytes(self._h, i, C.byref(b)))
            out[k] = b.value
        return out

    def prof_records(self, max_rows=16384):
        """numpy (n, 8): kind, ms, flops, bytes, M, N, K, tag per sampled launch, in launch order."""
        import numpy as np
        torch.cuda.synchronize(self.device)
        rows = np.zeros((max_rows, 8), dtype=np.float64)
        n = C.c_int()
        _lib.check(self._lib.hedit_prof_records(self._h, rows.ctypes.data_as(C.POINTER(C.c_double)), max_rows, C.byref(n)))
        return rows[:n.value]

    # ---------------------------------------------------------------- forward
    def forward_raw(self, sample, t, ctx, plan=None, out=None):
        """sample fp32 (B,C,H,W) cuda, t python float, ctx fp32 (B,77,D) cuda, plan: _lib.P2PPlan."""
        B, Cc, H, W = sample.shape
        if sample.dtype != torch.float32 or ctx.dtype != torch.float32:
            raise TypeError("sample and encoder_hidden_states must be float32")
        if ctx.shape != (B, 77, self.config["cross_attention_dim"]):
            raise ValueError(f"encoder_hidden_states must be ({B}, 77, {self.config['cross_attention_dim']}), "
                             f"got {tuple(ctx.shape)}")
        sample = sample.contiguous()
        ctx = ctx.contiguous()
        if out is None:
            out = torch.empty(B, self.config["out_channels"], H, W, dtype=torch.float32, device=sample.device)
        ws = self._workspace(B, H, W)
        _lib.check(self._lib.hedit_unet_forward(
            self._h, _lib.ptr(sample), C.c_float(float(t)), _lib.ptr(ctx), B, H, W,
            C.byref(plan) if plan is not None else None, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
            _lib.cur_stream()))
        return out

    _HOOK_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)

    def forward_hooked(self, sample, t, ctx, controller, save_attn=True, out=None):
        """One UNet call with a foreign (host-language) controller: ``controller(attention_probs, is_cross,
        place_in_unet, save_attn)`` is called once per attention layer, in execution order, with the reference's tensor
        (fp32 [batch * heads, queries, keys], a fresh tensor per layer like the reference's, so a controller may keep it);
        in-place changes go into the product with V, the return value is ignored (ptp_utils.py:100-106)."""
        B, _, H, W = sample.shape
        state = {"exc": None}
        places = ("down", "mid", "up")

        def cb(_user, probs, bh, nq, nk, is_cross, place, _layer, _stream):
            try:
                off = int(probs) - self._ws.data_ptr()
                view = self._ws[off:off + bh * nq * nk * 4].view(torch.float32).view(bh, nq, nk)
                attn = view.clone()
                controller(attn, bool(is_cross), places[place], save_attn)
                view.copy_(attn)
                return 0
            except BaseException as e:      # noqa: BLE001  (must not unwind through the C frames)
                state["exc"] = e
                return 1

        fn = self._HOOK_T(cb)
        _lib.check(self._lib.hedit_unet_set_attn_hook(self._h, C.cast(fn, C.c_void_p), None))
        try:
            try:
                return self.forward_raw(sample, t, ctx, None, out)      # (the workspace is sized with the hook set)
            except _lib.HipError:
                if state["exc"] is not None:
                    raise state["exc"]
                raise
        finally:
            _lib.check(self._lib.hedit_unet_set_attn_hook(self._h, None, None))

    def forward(self, sample, timestep=None, encoder_hidden_states=None, cross_attention_kwargs=None,
                return_dict=True):
        kw = dict(cross_attention_kwargs or {})
        use_controller = kw.pop("use_controller", True)
        save_attn = kw.pop("save_attn", True)
        use_editor = kw.pop("use_editor", True)          # MasaCtrl's switch (masactrl_utils.py:40)
        if kw:
            raise TypeError(f"unsupported cross_attention_kwargs {sorted(kw)}")
        if isinstance(timestep, torch.Tensor):
            tt = timestep.reshape(-1)
            if tt.numel() > 1 and not bool((tt == tt[0]).all()):
                raise NotImplementedError("per-sample timesteps: the batch must share one timestep")
            t = float(tt[0])
        else:
            t = float(timestep)
        sample = sample.to(device=self.device, dtype=torch.float32)
        ctx = encoder_hidden_states.to(device=self.device, dtype=torch.float32)
        controller = self._controller() if use_controller else None
        if controller is None and use_editor:
            # an attention editor registered on the UNet (MasaCtrl: regiter_attention_editor_diffusers; Plug-and-Play:
            # register_attention_control_efficient / register_conv_control_efficient)
            controller = getattr(self, "_attention_editor", None)
        from .p2p.ptp_classes import runs_in_python
        if runs_in_python(controller):
            return UNetOutput(sample=self.forward_hooked(sample, t, ctx, controller, save_attn))
        plan = None
        if controller is not None:
            plan = controller._plan(self, sample.shape[0], sample.shape[2], sample.shape[3], save_attn)
        eps = self.forward_raw(sample, t, ctx, plan)
        if controller is not None:
            controller._after_pass(save_attn)
        return UNetOutput(sample=eps)

    __call__ = forward
