"""The style image encoder of the combined text + style task -- drop-in for the reference's
text-guided-n-style/clip_guidance/base_clip.py:30-66 (``CLIPEncoder``) on the slice of
clip_guidance/clip/model.py it actually exercises.

``get_gram_matrix_residual`` reads ``feats[2]`` only: the token features after the THIRD residual
attention block of the CLIP ViT (base_clip.py:60-65, model.py:339-365), so this module holds the
patch embedding, ln_pre and the first ``n_blocks`` = 3 blocks; the other nine blocks, ln_post, the
projection and the text tower cannot influence the residual or its gradient and are not built.
Parameters carry the OpenAI CLIP state_dict names (``visual.conv1.weight`` ...), so a local
checkpoint of the reference's model (ViT-B/16) loads directly; nothing is ever downloaded.

The encoder runs natively and only natively (``hedit_vit_gram`` / ``hedit_vit_gram_fwd_bwd`` of libhedit_hip.so,
csrc/vit.hip): the Frobenius norm of the Gram residual AND its gradient w.r.t. the resized, normalised image come
from one call (fp32 token stream, split-bf16 contractions with fp32 accumulation -- finer than the reference's fp16
CLIP, model.py:414-435); the bicubic resize and the normalisation in front of it stay torch ops, so autograd carries
the gradient on into the VAE decoder's HIP backward (hedit.vae).  The torch modules below are PARAMETER CONTAINERS
(the OpenAI state_dict layout, no forward): there is no torch / CPU execution path in the product -- the fp32
restatement the golden vectors pin lives in oracle/reward_nets.py (test infrastructure)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
VIT_B16 = dict(width=768, layers=3, heads=12, patch_size=16, input_resolution=224)


class _Block(nn.Module):
    """Parameters of a ResidualAttentionBlock (model.py:167-190): x + MHA(LN(x)); x + MLP(LN(x)), QuickGELU."""

    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.Module()
        self.attn.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.attn.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.attn.out_proj = nn.Linear(d, d)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Module()
        self.mlp.c_fc = nn.Linear(d, 4 * d)
        self.mlp.c_proj = nn.Linear(4 * d, d)


class _Visual(nn.Module):
    def __init__(self, width, layers, heads, patch_size, input_resolution):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(width))
        self.positional_embedding = nn.Parameter(torch.zeros((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList(_Block(width, heads) for _ in range(layers))


class ClipVisualPrefix(nn.Module):
    """Parameters of conv1 -> [class; patches] + positional -> ln_pre -> the first ``layers`` blocks (model.py:339-357)."""

    def __init__(self, width=768, layers=3, heads=12, patch_size=16, input_resolution=224):
        super().__init__()
        self.input_resolution = input_resolution
        self.visual = _Visual(width, layers, heads, patch_size, input_resolution)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def load_clip_state_dict(self, sd):
        """Load from a full OpenAI-CLIP state_dict (extra keys -- later blocks, text tower -- are ignored)."""
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"CLIP state_dict lacks {missing[:4]} ({len(missing)} keys)")
        self.load_state_dict({k: sd[k].to(own[k].dtype) for k in own})
        return self

    def init_random(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("ln_1.weight") or name.endswith("ln_2.weight") or name.endswith("ln_pre.weight"):
                    p.fill_(1.0)
                elif p.dim() == 1 and "embedding" not in name:
                    p.zero_()
                else:
                    fan = p.shape[-1] if p.dim() == 2 else (p[0].numel() if p.dim() > 2 else p.shape[0])
                    p.copy_((torch.randn(p.shape, generator=g) * fan ** -0.5).to(p.dtype))
        return self


def read_clip_checkpoint(path):
    """A local OpenAI CLIP file: TorchScript archive (ViT-B-16.pt as the reference downloads it,
    base_clip.py:14-27) or a plain state_dict."""
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu")
        return sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()


_RESIZE_TABLES = {}


def _resize_tables(n_in, n_out, device):
    """torch's own bicubic weights (align_corners=False, A = -0.75, border clamping) as sparse tables: forward
    [n_out][4+] and transposed [n_in][*].  The axis map is extracted from F.interpolate on the identity, so the resize
    computes what the reference's F.interpolate(..., mode="bicubic") computes, tap for tap."""
    key = (n_in, n_out, str(device))
    if key not in _RESIZE_TABLES:
        w = F.interpolate(torch.eye(n_in, dtype=torch.float64)[None, None], size=(n_out, n_in), mode="bicubic")[0, 0]      # [n_out][n_in]

        def table(m):
            nnz = int((m != 0).sum(1).max())
            idx = torch.zeros(m.shape[0], nnz, dtype=torch.int32)
            val = torch.zeros(m.shape[0], nnz, dtype=torch.float32)
            for i in range(m.shape[0]):
                j = torch.nonzero(m[i]).flatten()
                idx[i, :len(j)] = j.int()
                val[i, :len(j)] = m[i, j].float()
            return idx.contiguous().to(device), val.contiguous().to(device), nnz
        _RESIZE_TABLES[key] = (table(w), table(w.t().contiguous()))
    return _RESIZE_TABLES[key]


def _axis_mix(x, tab, axis):
    from .. import _lib
    idx, val, nnz = tab
    x = x.float().contiguous()
    shape = list(x.shape)
    outer = 1
    for d in shape[:axis]:
        outer *= d
    inner = 1
    for d in shape[axis + 1:]:
        inner *= d
    n_out = idx.shape[0]
    out = torch.empty(shape[:axis] + [n_out] + shape[axis + 1:], device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().hedit_axis_mix(_lib.ptr(x), _lib.ptr(out), _lib.ptr(idx), _lib.ptr(val), nnz, outer, shape[axis], n_out, inner,
                                             _lib.cur_stream()))
    return out


class _BicubicResize(torch.autograd.Function):
    """F.interpolate(im, size=(S, S), mode="bicubic") with a deterministic, batch-invariant backward (hedit_axis_mix)"""

    @staticmethod
    def forward(ctx, im, size):
        H, W = im.shape[2], im.shape[3]
        ctx.dims = (H, W, size)
        fy, _ = _resize_tables(H, size, im.device)
        fx, _ = _resize_tables(W, size, im.device)
        return _axis_mix(_axis_mix(im.detach(), fy, 2), fx, 3)

    @staticmethod
    def backward(ctx, g):
        H, W, size = ctx.dims
        _, by = _resize_tables(H, size, g.device)
        _, bx = _resize_tables(W, size, g.device)
        return _axis_mix(_axis_mix(g, bx, 3), by, 2), None


class _NativeGramNorm(torch.autograd.Function):
    """sum_b |Gram(x_b) - Gram_ref|_F with the gradient w.r.t. x from the same native call"""

    @staticmethod
    def forward(ctx, x, owner, ref=None):
        loss, grad = owner._native_loss_and_grad(x.detach(), ref=ref)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.view(-1, 1, 1, 1), None, None


class _NativeGramResidual(torch.autograd.Function):
    """Gram(x_b) - Gram_ref as a differentiable (B, D, D) tensor, for callers that take the residual itself (the
    reference's closure: torch.linalg.norm(get_gram_matrix_residual(img)), h_edit.py:172-175).  Backward for an
    arbitrary upstream gradient U_b through the executor's norm-gradient entry: a Gram matrix is symmetric, so only the
    symmetric part of U_b acts on it; with the per-image reference R_b = Gram_b - s_b sym(U_b) (s_b > 0, see backward)
    that entry back-propagates (Gram_b - R_b) / |Gram_b - R_b| = sym(U_b) / |sym(U_b)|, so |sym(U_b)| times its result
    is exactly J^T U_b."""

    @staticmethod
    def forward(ctx, x, owner):
        xd = x.detach()
        gram = owner._native_gram(xd)
        ctx.owner = owner
        ctx.save_for_backward(xd, gram)
        return gram - owner._native_ref(x.device)

    @staticmethod
    def backward(ctx, up):
        xd, gram = ctx.saved_tensors
        up = 0.5 * (up.float() + up.float().transpose(1, 2))
        nrm = up.flatten(1).norm(dim=1)
        live = nrm > 0
        # The native entry normalises Gram - R, so any positive multiple of sym(U) gives the same direction: scale it
        # (by a power of two, exact) to the magnitude of the Gram entries before it is subtracted from them.  Unscaled,
        # Gram - (Gram - sym(U)) only returns sym(U) up to half an ulp of Gram: with ViT-B/16 features the Gram
        # entries are 1e3...1e5 (ulp 1e-4...1e-2) and the entries of a unit-norm U about 1e-3, i.e. mostly lost.
        gmax = gram.flatten(1).abs().amax(dim=1)
        umax = up.flatten(1).abs().amax(dim=1)
        s = torch.exp2(torch.round(torch.log2(torch.where(live & (gmax > 0), gmax / umax.clamp_min(1e-38), torch.ones_like(gmax)))))
        s = torch.where(torch.isfinite(s) & (s > 0), s, torch.ones_like(s)).view(-1, 1, 1)
        ref = torch.where(live.view(-1, 1, 1), gram - s * up, gram - 1.0).contiguous()  # (a dead row must not divide by 0)
        _, grad = ctx.owner._native_loss_and_grad(xd, ref=ref)
        return grad * torch.where(live, nrm, torch.zeros_like(nrm)).view(-1, 1, 1, 1), None


class CLIPEncoder(nn.Module):
    """Reference signature ``CLIPEncoder(need_ref=False, ref_path=None)`` plus where the weights come
    from (the reference downloads them; this build is offline): ``clip_path`` = local checkpoint,
    or ``clip_model`` = a ready ClipVisualPrefix, else seeded random weights (synthetic runs)."""

    def __init__(self, need_ref=False, ref_path=None, clip_path=None, clip_model=None, device=None,
                 dtype=torch.float16, seed=0):
        super().__init__()
        self._h = None
        self._ws = None
        self._gram_ref_native = None
        if clip_model is None:
            if clip_path is not None:
                sd = read_clip_checkpoint(clip_path)
                w = sd["visual.conv1.weight"]
                cfg = dict(width=w.shape[0], layers=3, heads=w.shape[0] // 64, patch_size=w.shape[-1],
                           input_resolution=w.shape[-1] * round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5))
                clip_model = ClipVisualPrefix(**cfg).load_clip_state_dict(sd)
            else:
                clip_model = ClipVisualPrefix(**VIT_B16).init_random(seed)
            clip_model = clip_model.to(dtype)
        self.clip_model = clip_model.eval()
        self.size = self.clip_model.input_resolution
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        # images arrive in [-1, 1]: Normalize(mean*2-1, std*2) of base_clip.py:38-41
        self.register_buffer("_mean", torch.tensor([m * 2 - 1 for m in CLIP_MEAN]).view(1, 3, 1, 1))
        self.register_buffer("_std", torch.tensor([s * 2 for s in CLIP_STD]).view(1, 3, 1, 1))
        if device is not None:
            self.to(device)
        if need_ref:
            self.set_reference(load_style_reference(ref_path, self.size).to(self._mean.device))

    def preprocess(self, im):
        return (im - self._mean.to(im.dtype)) / self._std.to(im.dtype)

    def set_reference(self, ref):
        """ref: (1, 3, S, S), already CLIP-normalised (base_clip.py:43-53).  Held as a (non-persistent) buffer so that
        ``CLIPEncoder(...).cuda()`` / ``.to()`` moves it with the module, as the reference's call pattern expects."""
        if "ref" in self._buffers:
            self.ref = ref
        else:
            self.register_buffer("ref", ref, persistent=False)
        self._gram_ref_native = None

    # ------------------------------------------------------------------ native executor (csrc/vit.hip)
    @staticmethod
    def _require_gpu(x):
        if not x.is_cuda:
            raise RuntimeError("CLIPEncoder runs on the HIP executor only: pass CUDA tensors (there is no CPU / torch path)")

    def _native(self, device):
        import ctypes as C
        from .. import _lib
        if self._h is not None:
            return self._h
        lib = _lib.lib()
        v = self.clip_model.visual
        cfg = _lib.VitCfg()
        cfg.width = v.conv1.weight.shape[0]
        cfg.layers = len(v.transformer.resblocks)
        cfg.heads = v.transformer.resblocks[0].heads
        cfg.patch_size = v.conv1.kernel_size[0]
        cfg.input_resolution = self.size
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.hedit_vit_create(C.byref(cfg), C.byref(h)))
            sd = self.clip_model.state_dict()
            for i in range(lib.hedit_vit_num_params(h)):
                name = lib.hedit_vit_param_name(h, i).decode()
                w = sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
                _lib.check(lib.hedit_vit_load(h, name.encode(), _lib.ptr(w), w.numel(), _lib.cur_stream()))
                torch.cuda.current_stream().synchronize()
            _lib.check(lib.hedit_vit_finalize(h, _lib.cur_stream()))
        self._h, self._lib, self._width = h, lib, cfg.width
        return h

    def __del__(self):
        if getattr(self, "_h", None) is not None and getattr(self, "_owns_handle", True):
            try:
                self._lib.hedit_vit_destroy(self._h)
            except Exception:
                pass

    def sibling(self, ref):
        """Another style reference on the SAME weights: shares this encoder's parameter container, native handle and
        workspace (a lock-step batch keeps one reference per image, not one copy of the network per image).
        ref: (1,3,S,S) CLIP-normalised tensor, or the path of a style image."""
        if isinstance(ref, str):
            ref = load_style_reference(ref, self.size)
        dev = self._mean.device
        if dev.type == "cuda":
            self._native(dev)
        sib = CLIPEncoder(clip_model=self.clip_model)
        sib.to(dev)
        sib.set_reference(ref.to(dev))
        if self._h is not None:
            sib._h, sib._lib, sib._width, sib._owns_handle, sib._parent = self._h, self._lib, self._width, False, self
        return sib

    @staticmethod
    def gram_residual_norms_each(encs, ims):
        """(N,3,H,W) -> (N,): image i against the style reference of encs[i], differentiable w.r.t. ``ims``.  Encoders
        that share one native handle (``sibling``) take ONE native call with one reference Gram matrix per image."""
        first = encs[0]
        first._require_gpu(ims)
        first._native(ims.device)
        if any(getattr(e, "_h", None) is None for e in encs[1:]) or any(e._h.value != first._h.value for e in encs[1:]):
            return torch.cat([e.gram_residual_norms(ims[i:i + 1]) for i, e in enumerate(encs)])
        x = first.preprocess(_BicubicResize.apply(ims, first.size))
        refs = torch.stack([e._native_ref(ims.device) for e in encs]).contiguous()
        return _NativeGramNorm.apply(x, first, refs)

    def _workspace(self, B, device):
        o = getattr(self, "_parent", None) or self          # siblings use their parent's workspace
        need = self._lib.hedit_vit_workspace_bytes(self._h, B)
        if o._ws is None or o._ws.numel() < need or o._ws.device != device:
            o._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return o._ws

    def __deepcopy__(self, memo):
        """a copy owns its own parameters and (lazily) its own native handle -- never a second reference to this one's"""
        import copy
        twin = CLIPEncoder(clip_model=copy.deepcopy(self.clip_model, memo))
        twin.to(self._mean.device)
        if "ref" in self._buffers:
            twin.set_reference(self.ref.detach().clone())
        return twin

    def _native_gram(self, x):
        """x (B,3,S,S) CLIP-normalised -> Gram matrices (B,D,D)"""
        from .. import _lib
        x = x.detach().float().contiguous()
        h = self._native(x.device)
        ws = self._workspace(x.shape[0], x.device)
        g = torch.empty(x.shape[0], self._width, self._width, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hedit_vit_gram(h, _lib.ptr(x), x.shape[0], _lib.ptr(g), _lib.ptr(ws), ws.numel(), _lib.cur_stream()))
        return g

    def _native_ref(self, device):
        if self._gram_ref_native is None or self._gram_ref_native.device != device:
            self._gram_ref_native = self._native_gram(self.ref.to(device))[0].contiguous()
        return self._gram_ref_native

    def _native_loss_and_grad(self, x, ref=None):
        """(loss [B] = |Gram(x_b) - Gram_ref|_F, d sum(loss) / d x) for x (B,3,S,S), CLIP-normalised; ``ref``: one
        reference Gram matrix per image (B,D,D) instead of this encoder's style reference"""
        from .. import _lib
        x = x.float().contiguous()
        B = x.shape[0]
        h = self._native(x.device)
        per_image = ref is not None
        if ref is None:
            ref = self._native_ref(x.device)
        ws = self._workspace(B, x.device)
        loss = torch.empty(B, device=x.device)
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.hedit_vit_gram_fwd_bwd(h, _lib.ptr(x), _lib.ptr(ref), int(per_image), B, 1.0, _lib.ptr(loss),
                                                        _lib.ptr(grad), _lib.ptr(ws), ws.numel(), _lib.cur_stream()))
        return loss, grad

    def gram_residual_norms(self, ims):
        """(N,3,H,W) in [-1, 1] -> (N,) = |get_gram_matrix_residual(ims[i:i+1])|_F, differentiable w.r.t. ``ims``: what the
        style closure needs (torch.linalg.norm of the residual, inversion/h_edit.py:172-175).  One native call evaluates
        norm and gradient; resize + normalisation stay torch ops in front of it."""
        self._require_gpu(ims)
        x = self.preprocess(_BicubicResize.apply(ims, self.size))
        return _NativeGramNorm.apply(x, self)

    def get_gram_matrix_residual(self, im1):
        """Gram matrix (D x D) of the block-3 patch tokens of ``im1`` (batch item 0, as the reference) minus that of the
        style reference (base_clip.py:55-66), differentiable w.r.t. ``im1``.  The reference's Gram matrix does not depend
        on ``im1``; it is computed on first use and kept (the reference recomputes it every call with the same result)."""
        return self.gram_residuals(im1[:1])[0]

    def gram_residuals(self, ims):
        """Batched form for images that share this style reference: (N, 3, H, W) -> (N, D, D), row i equal to
        get_gram_matrix_residual(ims[i:i+1]) (one encoder pass for the N images of a lock-step batch)."""
        self._require_gpu(ims)
        x = self.preprocess(_BicubicResize.apply(ims, self.size))
        return _NativeGramResidual.apply(x, self)


def load_style_reference(path, size=224):
    """PIL RGB -> bilinear resize to size x size -> [0,1] CHW -> Normalize(CLIP mean, std) -> (1,3,S,S)
    (base_clip.py:43-53: torchvision ToTensor + Normalize restated with numpy)."""
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size, size), Image.Resampling.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    return ((x - mean) / std).unsqueeze(0)
