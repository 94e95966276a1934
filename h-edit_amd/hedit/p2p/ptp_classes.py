"""Prompt-to-Prompt controllers for the HIP path.

Class names, constructor arguments and public attributes follow the reference's
text-guided/p2p/ptp_classes.py (LocalBlend :17-72, AttentionControl :74-118, AttentionStore
:124-160, AttentionControlEdit :162-227, AttentionReplace :229-243, AttentionRefine :245-262,
AttentionReweight :264-283, get_equalizer :285-294, load_512 :351-373), so driver code written
for the reference runs unchanged.  What differs is WHERE the edit executes: instead of mutating a
materialised probability tensor from Python 32 times per UNet call, a controller compiles, once,
the per-step tables

    P_new = P_src . A_s + bvec_s * P_tar          (A_s: 77x77, bvec_s: 77, s = cur_step)

that encode Replace / Refine / Reweight plus the cross_replace_alpha blend, and hands the HIP
cross-attention kernel a pointer into them; self-attention replacement becomes "the target row
reads the source row's Q and K".  The attention store keeps the cross maps only (the reference
also keeps <=32x32 self maps but nothing on the h-Edit path reads them) and is accumulated
in-kernel on save_attn passes.

A controller edits ONE (source, target) prompt pair, exactly like the reference;
``ControllerBatch`` stacks several for image-batched sampling (build-side generalisation,
SURVEY.md section 0.1).
"""
import abc
import ctypes as C

import numpy as np
import torch

from .. import _lib
from . import ptp_utils, seq_aligner

LOW_RESOURCE = False
MAX_NUM_WORDS = 77
WPAD = 96


class LocalBlend:
    def __init__(self, prompts, num_steps, words, substruct_words=None, start_blend=0.2, th=(.3, .3),
                 tokenizer=None, device=None):
        self.max_num_words = MAX_NUM_WORDS
        alpha_layers = torch.zeros(len(prompts), 1, 1, 1, 1, self.max_num_words)
        for i, (prompt, words_) in enumerate(zip(prompts, words)):
            if isinstance(words_, str):
                words_ = [words_]
            for word in words_:
                ind = ptp_utils.get_word_inds(prompt, word, tokenizer)
                alpha_layers[i, :, :, :, :, ind] = 1
        self.substruct_layers = None
        if substruct_words is not None:          # ptp_classes.py:28-38: a second word mask, cut out of the blend mask
            sub = torch.zeros(len(prompts), 1, 1, 1, 1, self.max_num_words)
            for i, (prompt, words_) in enumerate(zip(prompts, substruct_words)):
                if isinstance(words_, str):
                    words_ = [words_]
                for word in words_:
                    ind = ptp_utils.get_word_inds(prompt, word, tokenizer)
                    sub[i, :, :, :, :, ind] = 1
            self.substruct_layers = sub.to(device) if device is not None else sub
        self.alpha_layers = alpha_layers.to(device) if device is not None else alpha_layers
        self.start_blend = int(start_blend * num_steps)
        self.counter = 0
        self.th = th

    def __call__(self, x_t, attention_store):
        self.counter += 1
        if self.counter > self.start_blend:
            maps = attention_store["down_cross"][2:4] + attention_store["up_cross"][:3]
            _blend_launch(x_t, maps, [self], 1)
        return x_t


def _blend_launch(x_t, maps, blends, n_img):
    """x_t: (2*n_img, C, H, W) fp32 cuda laid out [x_orig * n, x_edit * n]; edited in place."""
    lib = _lib.lib()
    if x_t.dtype != torch.float32 or not x_t.is_cuda or not x_t.is_contiguous():
        raise TypeError("LocalBlend expects a contiguous float32 CUDA latent")
    for m in maps:
        if m.shape[-2] != 256:
            raise ValueError("LocalBlend reads 16x16 cross maps (reference hard-codes 16x16: "
                             "ptp_classes.py:59-62)")
    heads = maps[0].numel() // (n_img * 2 * 256 * MAX_NUM_WORDS)
    alpha = torch.zeros(n_img, 2, MAX_NUM_WORDS)
    sub = torch.zeros(n_img, 2, MAX_NUM_WORDS)
    enabled = torch.zeros(n_img, dtype=torch.int32)
    th0 = th1 = None
    has_sub = False
    for i, lb in enumerate(blends):
        if lb is not None:
            alpha[i] = lb.alpha_layers.reshape(2, MAX_NUM_WORDS).cpu()
            if lb.substruct_layers is not None:
                sub[i] = lb.substruct_layers.reshape(2, MAX_NUM_WORDS).cpu()
                has_sub = True
            enabled[i] = 1
            if th0 is not None and float(lb.th[0]) != th0:
                raise ValueError("LocalBlend thresholds differ inside one lock-step batch (the blend kernel takes one value)")
            th0 = float(lb.th[0])
            if lb.substruct_layers is not None:      # th[1] is read for images with substruct words only
                if th1 is not None and float(lb.th[1]) != th1:
                    raise ValueError("LocalBlend substruct thresholds differ inside one lock-step batch")
                th1 = float(lb.th[1])
    th = (0.3 if th0 is None else th0, 0.3 if th1 is None else th1)
    alpha = alpha.to(x_t.device)
    enabled = enabled.to(x_t.device)
    arr = (C.c_void_p * len(maps))(*[m.data_ptr() for m in maps])
    _, Cc, H, W = x_t.shape
    if has_sub:
        # images of the batch without substruct words carry an all-zero layer: their mean map is 0, 0 / 0 is never > th
        sub = sub.to(x_t.device)
        _lib.check(lib.hedit_local_blend_sub(arr, len(maps), heads, _lib.ptr(alpha), _lib.ptr(sub), _lib.ptr(enabled),
                                             _lib.ptr(x_t), n_img, Cc, H, W, C.c_float(th[0]), C.c_float(th[1]),
                                             _lib.cur_stream()))
    else:
        _lib.check(lib.hedit_local_blend(arr, len(maps), heads, _lib.ptr(alpha), _lib.ptr(enabled),
                                         _lib.ptr(x_t), n_img, Cc, H, W, C.c_float(th[0]), _lib.cur_stream()))
    # alpha / sub / enabled are consumed by a stream-ordered kernel: keep them alive on the tensor
    x_t._hedit_keep = (alpha, sub, enabled)


class AttentionControl(abc.ABC):
    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    @property
    def num_uncond_att_layers(self):
        return self.num_att_layers if LOW_RESOURCE else 0

    # ---- the reference's Python protocol (ptp_classes.py:81-108), for controllers that ARE Python: a subclass that
    # overrides forward() -- the reference's extension point -- has no compiled edit tables, so the UNet hands it the
    # materialised probabilities of every layer through the executor's hook (hedit/unet.py::forward_hooked) and this
    # __call__ does the reference's slicing and counting.  hedit's own controllers never define forward(): their edit
    # runs inside the attention kernels (_plan below).
    def forward(self, attn, is_cross, place_in_unet, save_attn):
        raise NotImplementedError

    def __call__(self, attn, is_cross, place_in_unet, save_attn):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            h = attn.shape[0]
            attn[h // 2:] = self.forward(attn[h // 2:], is_cross, place_in_unet, save_attn)    # the conditional half only
        if not save_attn:
            return attn
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()
        return attn

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    # ---- protocol with hedit.unet.UNet2DConditionModel
    def _after_pass(self, save_attn):
        """What the reference's per-layer ``__call__`` bookkeeping amounts to after one full UNet
        pass over num_att_layers layers (ptp_classes.py:100-108)."""
        if not save_attn:
            return
        self.cur_att_layer = 0
        self.cur_step += 1
        self.between_steps()

    def _plan(self, unet, B, H, W, save_attn):
        return None


class EmptyControl(AttentionControl):
    pass


class _PlanState:
    """Device-side tables + store buffers for a list of per-image controllers."""

    def __init__(self, members, unet, B, H, W):
        n = len(members)
        if B < 4 * n:
            raise ValueError(f"a P2P pass over {n} image(s) expects a batch of {4 * n} rows laid out "
                             f"[x_orig|null]*n, [x_edit|null]*n, [x_orig|src]*n, [x_edit|tar]*n "
                             f"(+ optional extra un-edited rows); got {B}")
        dev = unet.device
        self.key = (id(unet), B, H, W)
        self.n = n
        ar = torch.arange(B, dtype=torch.int32)
        self.pair_src = ar[2 * n:3 * n].contiguous().to(dev)
        self.pair_tar = ar[3 * n:4 * n].contiguous().to(dev)
        # rows outside the conditional (src,tar) quarter pairs are never edited: the unconditional
        # half and any extra rows appended after the 4n block (e.g. the source-prompt pass)
        self.singles = torch.cat([ar[:2 * n], ar[4 * n:]]).contiguous().to(dev)
        self.n_single = int(self.singles.numel())
        qk = ar.clone()
        qk[3 * n:4 * n] = ar[2 * n:3 * n]
        self.qk_src = qk.to(dev)
        T1 = members[0]._n_table_steps()
        mixT = torch.zeros(T1, n, WPAD, WPAD)
        bvec = torch.zeros(T1, n, WPAD)
        for i, m in enumerate(members):
            A, b = m._mix_tables()                      # (T1,77,77), (T1,77)
            mixT[:, i, :MAX_NUM_WORDS, :MAX_NUM_WORDS] = A.transpose(1, 2)
            bvec[:, i, :MAX_NUM_WORDS] = b
        self.mixT = mixT.to(_lib.storage_dtype()).contiguous().to(dev)
        self.bvec = bvec.contiguous().to(dev)
        self.layers = unet.store_layers(H, W)
        heads = unet.heads
        self.bufs = [torch.zeros(n, 2, heads, tok, MAX_NUM_WORDS, dtype=torch.float32, device=dev)
                     for tok, _ in self.layers]
        self.h_store = (C.c_void_p * len(self.bufs))(*[b.data_ptr() for b in self.bufs])
        store = {f"{p}_{k}": [] for k in ("cross", "self") for p in ("down", "mid", "up")}
        for (tok, place), buf in zip(self.layers, self.bufs):
            view = buf.reshape(2 * heads, tok, MAX_NUM_WORDS) if n == 1 else buf.reshape(n, 2 * heads, tok, MAX_NUM_WORDS)
            store[f"{place}_cross"].append(view)
        # the <= 32 x 32 SELF maps the reference's store also keeps (ptp_classes.py:135-150): nothing on the h-Edit path reads
        # them and they are 67 MB per image and 32 x 32 layer, so they are kept on request only (store_self_maps)
        self.self_bufs, self.h_store_self = [], None
        if any(getattr(m, "store_self_maps", False) for m in members):
            self.self_bufs = [torch.zeros(n, 2, heads, tok, tok, dtype=torch.float32, device=dev) for tok, _ in self.layers]
            self.h_store_self = (C.c_void_p * len(self.self_bufs))(*[b.data_ptr() for b in self.self_bufs])
            for (tok, place), buf in zip(self.layers, self.self_bufs):
                store[f"{place}_self"].append(buf.reshape(2 * heads, tok, tok) if n == 1 else buf.reshape(n, 2 * heads, tok, tok))
        self.attention_store = store

    def plan(self, cur_step, self_window, save_attn, edit=True):
        p = _lib.P2PPlan()
        p.mode = 2 if save_attn else 1
        p.n_pairs = self.n
        p.pair_src = self.pair_src.data_ptr()
        p.pair_tar = self.pair_tar.data_ptr()
        p.singles = self.singles.data_ptr()
        p.n_single = self.n_single
        in_window = edit and self_window[0] <= cur_step < self_window[1]
        p.qk_src = self.qk_src.data_ptr() if in_window else None
        s = min(cur_step, self.mixT.shape[0] - 1)
        p.mixT = self.mixT[s].data_ptr()
        p.bvec = self.bvec[s].data_ptr()
        p.h_store = C.cast(self.h_store, C.POINTER(C.c_void_p))
        p.n_store = len(self.bufs)
        if self.h_store_self is not None:
            p.h_store_self = C.cast(self.h_store_self, C.POINTER(C.c_void_p))
            p.n_store_self = len(self.self_bufs)
        return p


class AttentionStore(AttentionControl):
    # hedit extension: True = the fused path also keeps the <= 32 x 32 SELF maps ("down_self" / "mid_self" / "up_self"), as the
    # reference's store does (ptp_classes.py:135-150).  Off by default: no driver reads them, 67 MB per image and 32 x 32 layer.
    # Set on the instance (or the class) before the first UNet pass.
    store_self_maps = False

    def __init__(self, store_self_maps=None):
        super().__init__()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
        self._state = None
        if store_self_maps is not None:
            self.store_self_maps = bool(store_self_maps)

    @staticmethod
    def get_empty_store():
        return {"down_cross": [], "mid_cross": [], "up_cross": [],
                "down_self": [], "mid_self": [], "up_self": []}

    def get_average_attention(self):
        return {k: [item / self.cur_step for item in v] for k, v in self.attention_store.items()}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
        if self._state is not None:
            for b in self._state.bufs + self._state.self_bufs:
                b.zero_()

    # ---- the reference's Python protocol, for SUBCLASSES that override forward() and call super().forward(...) (the
    # reference's extension idiom, ptp_classes.py:135-150): such a controller runs on the hook path, where nothing
    # compiles a plan, so the store has to be kept here.  On the fused path forward() is never called and step_store
    # stays empty (the kernels accumulate into _PlanState.bufs), which makes between_steps() a no-op there.
    def forward(self, attn, is_cross, place_in_unet, save_attn=True):
        if save_attn and attn.shape[1] <= 32 ** 2:
            # the tensor itself, not a copy: an edit applied to it afterwards is what gets accumulated, as in the reference
            self.step_store[f"{place_in_unet}_{'cross' if is_cross else 'self'}"].append(attn)
        return attn

    def between_steps(self):
        if not any(self.step_store.values()):
            return
        if not self.attention_store:
            self.attention_store = self.step_store
        else:
            for key, maps in self.step_store.items():
                acc = self.attention_store[key]
                for i, m in enumerate(maps):
                    acc[i] += m
        self.step_store = self.get_empty_store()

    # ---- plan protocol
    def _members(self):
        return [self]

    def _n_table_steps(self):
        return 1

    def _mix_tables(self):
        """store-only controller: identity edit (P_new = P_tar)."""
        return torch.zeros(1, MAX_NUM_WORDS, MAX_NUM_WORDS), torch.ones(1, MAX_NUM_WORDS)

    def _self_window(self):
        return (0, 0)

    def _plan(self, unet, B, H, W, save_attn):
        members = self._members()
        if self._state is None or self._state.key != (id(unet), B, H, W):
            # a pass of another shape (4n rows, then 5n rows; another resolution): new device buffers -- the store must
            # follow them, maps accumulated in the old buffers would be read by LocalBlend otherwise
            self._state = _PlanState(members, unet, B, H, W)
            self.attention_store = {}
        if save_attn and not self.attention_store:
            self.attention_store = self._state.attention_store
        self._live_plan = self._state.plan(self.cur_step, self._self_window(), save_attn)
        return self._live_plan


class AttentionControlEdit(AttentionStore, abc.ABC):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend,
                 tokenizer, device):
        super().__init__()
        self.tokenizer = tokenizer
        self.device = device
        self.batch_size = len(prompts)
        if self.batch_size != 2:
            raise NotImplementedError("one (source, target) prompt pair per controller, as in the "
                                      "h-Edit drivers; stack controllers with ControllerBatch")
        self.cross_replace_alpha = ptp_utils.get_time_words_attention_alpha(
            prompts, num_steps, cross_replace_steps, self.tokenizer)
        if isinstance(self_replace_steps, float):
            self_replace_steps = 0, self_replace_steps
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.local_blend = local_blend

    @abc.abstractmethod
    def replace_cross_attention(self, attn_base, att_replace):
        """Same contract as the reference (attn_base (h,p,77), att_replace (1,h,p,77)); used on
        the host, on basis vectors, to derive the mixing tables."""
        raise NotImplementedError

    def step_callback(self, x_t):
        if self.local_blend is not None:
            x_t = self.local_blend(x_t, self.attention_store)
        return x_t

    def forward(self, attn, is_cross, place_in_unet, save_attn):
        """The reference's per-layer edit on a materialised map (ptp_classes.py:202-227), for subclasses that extend it
        through super().forward(...) on the hook path.  attn: the conditional half [source heads | target heads] of one
        layer, edited in place."""
        super().forward(attn, is_cross, place_in_unet, save_attn)
        lo, hi = self.num_self_replace
        if not is_cross and not (lo <= self.cur_step < hi):
            return attn
        heads = attn.shape[0] // self.batch_size
        src, tar = attn[:heads], attn[heads:]
        if is_cross:
            a = self.cross_replace_alpha[self.cur_step].to(device=attn.device, dtype=attn.dtype)[0]       # (1, 1, 77)
            mixed = self.replace_cross_attention(src, tar[None])[0]
            tar.copy_(mixed * a + (1 - a) * tar)
        elif attn.shape[1] <= 32 ** 2:               # self maps above 32 x 32 keep the target's own attention
            tar.copy_(src)
        return attn

    def _self_window(self):
        return self.num_self_replace

    def _n_table_steps(self):
        return self.cross_replace_alpha.shape[0]

    def _mix_tables(self):
        """Probe replace_cross_attention (linear in attn_base, diagonal in att_replace) with basis
        inputs, then fold in the per-step alpha blend of AttentionControlEdit.forward
        (ptp_classes.py:215-220):  new = R(base, repl) * a_s + (1 - a_s) * repl."""
        W = MAX_NUM_WORDS
        eye = torch.eye(W).reshape(1, W, W)                         # h=1, p=w index, n
        zero = torch.zeros(1, 1, W, W)
        A = self.replace_cross_attention(eye, zero).reshape(W, W)   # A[w][n]
        one = torch.ones(1, 1, 1, W)
        b = self.replace_cross_attention(torch.zeros(1, 1, W), one).reshape(W)
        alpha = self.cross_replace_alpha[:, 0, 0, 0, :].float().cpu()      # (T+1, 77)
        A_s = A[None] * alpha[:, None, :]
        b_s = b[None] * alpha + (1 - alpha)
        return A_s, b_s


def _like(table, ref):
    """A host-side table on the device / dtype of the map it is applied to (the tables live on the CPU: the fused path only
    ever reads them there, the hook path applies them to device tensors)."""
    return table.to(device=ref.device, dtype=ref.dtype)


class AttentionReplace(AttentionControlEdit):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend=None,
                 tokenizer=None, device=None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer, device)
        self.mapper = seq_aligner.get_replacement_mapper(prompts, self.tokenizer)

    def replace_cross_attention(self, attn_base, att_replace):
        return torch.einsum("hpw,bwn->bhpn", attn_base, _like(self.mapper, attn_base))


class AttentionRefine(AttentionControlEdit):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend=None,
                 tokenizer=None, device=None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer, device)
        self.mapper, alphas = seq_aligner.get_refinement_mapper(prompts, self.tokenizer)
        self.alphas = alphas.reshape(alphas.shape[0], 1, 1, alphas.shape[1])

    def replace_cross_attention(self, attn_base, att_replace):
        base = attn_base[:, :, self.mapper.to(attn_base.device)].permute(2, 0, 1, 3)
        al = _like(self.alphas, attn_base)
        return base * al + att_replace * (1 - al)


class AttentionReweight(AttentionControlEdit):
    def __init__(self, prompts, num_steps, cross_replace_steps, self_replace_steps, equalizer,
                 local_blend=None, controller=None, tokenizer=None, device=None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend, tokenizer, device)
        self.equalizer = equalizer
        self.prev_controller = controller

    def replace_cross_attention(self, attn_base, att_replace):
        eq = _like(self.equalizer, attn_base)[:, None, None, :]
        if self.prev_controller is not None:
            return self.prev_controller.replace_cross_attention(attn_base, att_replace) * eq
        return attn_base[None, :, :, :] * eq


class ControllerBatch(AttentionStore):
    """Several single-pair controllers driven in lock-step over an image batch.
    UNet batch layout: [x_orig|null]*n, [x_edit|null]*n, [x_orig|src]*n, [x_edit|tar]*n."""

    def __init__(self, controllers):
        super().__init__()
        if not controllers:
            raise ValueError("empty controller batch")
        self.controllers = list(controllers)
        win = {c._self_window() for c in self.controllers}
        steps = {c._n_table_steps() for c in self.controllers}
        if len(win) != 1 or len(steps) != 1:
            raise ValueError("controllers of one batch must share num_steps and self_replace_steps")

    @property
    def n_images(self):
        return len(self.controllers)

    def _members(self):
        return self.controllers

    def _self_window(self):
        return self.controllers[0]._self_window()

    def _n_table_steps(self):
        return self.controllers[0]._n_table_steps()

    def step_callback(self, x_t):
        blends = [getattr(c, "local_blend", None) for c in self.controllers]
        live = [b for b in blends if b is not None]
        if not live:
            return x_t
        for b in live:
            b.counter += 1
        if live[0].counter > live[0].start_blend:
            maps = self.attention_store["down_cross"][2:4] + self.attention_store["up_cross"][:3]
            _blend_launch(x_t, maps, blends, len(blends))
        return x_t


def get_equalizer(text, word_select, values, tokenizer):
    if isinstance(word_select, (int, str)):
        word_select = (word_select,)
    equalizer = torch.ones(len(values), 77)
    values = torch.tensor(values, dtype=torch.float32)
    for word in word_select:
        inds = ptp_utils.get_word_inds(text, word, tokenizer)
        equalizer[:, inds] = values
    return equalizer


from ..utils.utils import load_512  # noqa: E402,F401  (reference: p2p/ptp_classes.py:351-373)


def runs_in_python(controller):
    """True for a controller the UNet has to call layer by layer on materialised probabilities: anything that is not one
    of hedit's (no _plan), and a subclass of AttentionControl that overrides the reference's forward()."""
    if controller is None:
        return False
    if not hasattr(controller, "_plan"):
        return True
    if not isinstance(controller, AttentionControl):
        return False
    # hedit's own classes keep the reference's forward() only as the base of such subclasses; their edit runs in-kernel
    own = {AttentionControl.forward, AttentionStore.forward, AttentionControlEdit.forward}
    return getattr(type(controller), "forward", AttentionControl.forward) not in own
