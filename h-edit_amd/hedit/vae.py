"""AutoencoderKL facade over the native HIP executor (csrc/vae.hip).

The surface the reference uses of diffusers' VAE:
``vae.encode(image).latent_dist.mode()`` (text-guided/main_p2p.py:159, p2p/ptp_classes.py:351-373)
and ``vae.decode(1 / 0.18215 * latents).sample`` (main_p2p.py:263); ``config.scaling_factor``.
Weights are addressed by their diffusers state_dict names (the pre-0.18 attention names
query/key/value/proj_attn are accepted as aliases).  GPU only -- there is no eager path.
"""
import ctypes as C

import torch

from . import _lib
from .unet import random_state_dict

SD15_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
TINY_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(64, 128),
                       layers_per_block=1, norm_num_groups=32, scaling_factor=0.18215)

_OLD_ATTN_NAMES = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}


class DecoderOutput(dict):
    @property
    def sample(self):
        return self["sample"]


class _LatentDist:
    """Only the part of DiagonalGaussianDistribution the reference reads: ``.mode()`` / ``.mean``."""

    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        raise NotImplementedError("only latent_dist.mode() is provided (what the reference calls); "
                                  "the log-variance half of the moments is not computed")


class EncoderOutput:
    def __init__(self, mean):
        self.latent_dist = _LatentDist(mean)


class _Config(dict):
    __getattr__ = dict.__getitem__


class _DecodeFn(torch.autograd.Function):
    """``vae.decode`` as a differentiable node: the reference's style-guidance closure calls
    ``torch.autograd.grad(loss, latents)`` through ``model.vae.decode``
    (text-guided-n-style/inversion/h_edit.py:146-185).  The backward is one C call
    (hedit_vae_decode_vjp: forward with kept activations + input-gradient pass, all HIP); only the
    gradient w.r.t. the latents exists -- the weights are constants here as in the reference.
    The forward leaves its tape in a workspace of the VAE object (hedit_vae_decode_keep) and the
    backward consumes it (hedit_vae_decode_backward); if another differentiable decode of the same
    VAE ran in between, the backward falls back to the one-call form (forward again + backward)."""

    @staticmethod
    def forward(ctx, z, vae):
        ctx.vae = vae
        ctx.save_for_backward(z)
        img, ctx.tape_id = vae._decode_keep(z)
        return img

    @staticmethod
    def backward(ctx, d_image):
        (z,) = ctx.saved_tensors
        vae = ctx.vae
        if vae._tape_id == ctx.tape_id and vae._tape_live:
            return vae._decode_backward(z, d_image), None
        return vae.decode_vjp(z, d_image), None


class AutoencoderKL:
    def __init__(self, config=None, device="cuda:0"):
        cfg = dict(SD15_VAE_CONFIG)
        cfg.update(config or {})
        self.config = _Config(cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hedit.AutoencoderKL runs on the GPU only (HIP kernels)")
        self._lib = _lib.lib()
        c = _lib.VaeCfg()
        c.in_channels, c.latent_channels = cfg["in_channels"], cfg["latent_channels"]
        ch = list(cfg["block_out_channels"])
        c.n_levels = len(ch)
        for i, v in enumerate(ch):
            c.block_out_channels[i] = v
        c.layers_per_block, c.norm_num_groups = cfg["layers_per_block"], cfg["norm_num_groups"]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_vae_create(C.byref(c), C.byref(h)))
        self._h = h
        self.factor = 2 ** (len(ch) - 1)
        self.param_shapes = {}
        nd, dims = C.c_int(), (C.c_int * 4)()
        for i in range(self._lib.hedit_vae_num_params(self._h)):
            name = self._lib.hedit_vae_param_name(self._h, i).decode()
            _lib.check(self._lib.hedit_vae_param_shape(self._h, i, C.byref(nd), dims))
            self.param_shapes[name] = tuple(dims[k] for k in range(nd.value))
        self._ws = None
        self._ws_tape = None          # workspace holding the tape of the last differentiable decode
        self._tape_id, self._tape_live = 0, False

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.hedit_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    @staticmethod
    def current_names(sd):
        """pre-0.18 attention parameter names (query / key / value / proj_attn) -> the current ones"""
        ren = {}
        for k, v in sd.items():
            for old, new in _OLD_ATTN_NAMES.items():
                if old in k and ".attentions." in k:
                    k = k.replace(old, new)
            ren[k] = v
        return ren

    def load_state_dict(self, sd, strict=True):
        sd = self.current_names(sd)
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
                if w.dim() == 2 and len(shape) == 4:       # pre-0.18 checkpoints store 1x1 convs as linears and vice versa
                    w = w[:, :, None, None]
                if w.dim() == 4 and len(shape) == 2:
                    w = w[:, :, 0, 0]
                if tuple(w.shape) != shape:
                    raise ValueError(f"{name}: expected shape {shape}, got {tuple(w.shape)}")
                w = w.detach().to(device=self.device, dtype=torch.float32).contiguous()
                _lib.check(self._lib.hedit_vae_load(self._h, name.encode(), _lib.ptr(w), w.numel(), st))
                torch.cuda.current_stream().synchronize()
        return self

    def init_random(self, seed=0):
        sd = random_state_dict(self.param_shapes, seed)
        self.load_state_dict(sd)
        return sd

    # ---------------------------------------------------------------- passes
    def _workspace(self, B, lh, lw, encode):
        need = self._lib.hedit_vae_workspace_bytes(self._h, B, lh, lw, int(encode))
        if need == 0:
            raise RuntimeError("hedit_vae_workspace_bytes failed: " + self._lib.hedit_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def decode(self, z, return_dict=True):
        """z (B, latent_channels, h, w) -> .sample (B, in_channels, h*f, w*f), fp32.  Differentiable
        w.r.t. z when z requires grad (see _DecodeFn)."""
        if torch.is_grad_enabled() and z.requires_grad:
            img = _DecodeFn.apply(z.to(device=self.device, dtype=torch.float32), self)
        else:
            img = self._decode_raw(z)
        return DecoderOutput(sample=img) if return_dict else (img,)

    def decode_vjp(self, z, d_image):
        """d_z = (d decode(z) / d z)^T d_image, fp32 like z."""
        z = z.detach().to(device=self.device, dtype=torch.float32).contiguous()
        d_image = d_image.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B, _, lh, lw = z.shape
        if tuple(d_image.shape) != (B, self.config["in_channels"], lh * self.factor, lw * self.factor):
            raise ValueError(f"d_image shape {tuple(d_image.shape)} does not match decode({tuple(z.shape)})")
        ws = self._workspace(B, lh, lw, 2)
        dz = torch.empty_like(z)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_vae_decode_vjp(self._h, _lib.ptr(z), _lib.ptr(d_image), B, lh, lw, _lib.ptr(dz),
                                                      None, _lib.ptr(ws), ws.numel(), _lib.cur_stream()))
        return dz

    def _decode_keep(self, z):
        z = z.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B, _, lh, lw = z.shape
        need = self._lib.hedit_vae_workspace_bytes(self._h, B, lh, lw, 2)
        if need == 0:
            raise RuntimeError("hedit_vae_workspace_bytes failed: " + self._lib.hedit_last_error().decode())
        if self._ws_tape is None or self._ws_tape.numel() < need:
            self._ws_tape = None
            self._ws_tape = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws_tape
        img = torch.empty(B, self.config["in_channels"], lh * self.factor, lw * self.factor, dtype=torch.float32,
                          device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_vae_decode_keep(self._h, _lib.ptr(z), B, lh, lw, _lib.ptr(img), _lib.ptr(ws),
                                                       ws.numel(), _lib.cur_stream()))
        self._tape_id += 1
        self._tape_live = True
        return img, self._tape_id

    def _decode_backward(self, z, d_image):
        d_image = d_image.detach().to(device=self.device, dtype=torch.float32).contiguous()
        dz = torch.empty(z.shape, dtype=torch.float32, device=self.device)
        self._tape_live = False
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_vae_decode_backward(self._h, _lib.ptr(d_image), _lib.ptr(dz),
                                                           _lib.ptr(self._ws_tape), _lib.cur_stream()))
        return dz

    def _decode_raw(self, z):
        z = z.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B, _, lh, lw = z.shape
        ws = self._workspace(B, lh, lw, False)
        img = torch.empty(B, self.config["in_channels"], lh * self.factor, lw * self.factor, dtype=torch.float32,
                          device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_vae_decode(self._h, _lib.ptr(z), B, lh, lw, _lib.ptr(img), _lib.ptr(ws),
                                                  ws.numel(), _lib.cur_stream()))
        return img

    def encode(self, x):
        """x (B, in_channels, H, W) in [-1, 1] -> EncoderOutput with latent_dist.mode() (B, latent_channels, H/f, W/f)."""
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        if H % self.factor or W % self.factor:
            raise ValueError(f"image size must be a multiple of {self.factor}")
        lh, lw = H // self.factor, W // self.factor
        ws = self._workspace(B, lh, lw, True)
        mean = torch.empty(B, self.config["latent_channels"], lh, lw, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hedit_vae_encode(self._h, _lib.ptr(x), B, H, W, _lib.ptr(mean), _lib.ptr(ws),
                                                  ws.numel(), _lib.cur_stream()))
        return EncoderOutput(mean)
