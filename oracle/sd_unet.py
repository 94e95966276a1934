"""SD-1.x eps-network, CPU fp32 eager (oracle; see oracle/__init__.py).

The reference obtains this network from the third-party package diffusers==0.18.0
(``StableDiffusionPipeline.from_pretrained`` at text-guided/main_p2p.py:106; every call site
``model.unet(...)`` in text-guided/inversion/p2p_h_edit.py:98,123,245,281,315,458,484,492,613,
644,652 and ddpm_inversion.py:130,132).  diffusers is NOT under /root/reference and not in this
image, and the reference holds no test/golden vector for it, so the network body below is a
restatement of the PUBLISHED architecture (UNet2DConditionModel defaults used by SD-1.4/1.5,
SURVEY.md appendix A.7) => PARITY UNPINNED for the body.  What IS anchored on reference code:
  * attention order of operations: the module exposes the ``Attention`` surface that the
    reference's own processor consumes (text-guided/p2p/ptp_utils.py:65-122; pinned by g6);
  * ResBlock order of operations: same order as the in-tree restatement
    text-guided/plug_n_play/pnp_utils.py:96-150 (norm1, act, conv1, +temb proj, norm2, act,
    conv2, shortcut, add);
  * call/registry surface: SURVEY.md §8b (``.sample``/``["sample"]``, ``attn_processors``,
    ``set_attn_processor``, ``in_channels``, ``sample_size``);
  * parameter inventory: state_dict keys are diffusers' names; SD-1.x config gives 859.5 M params.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15 = dict(in_channels=4, out_channels=4, sample_size=64,
            block_out_channels=(320, 640, 1280, 1280),
            down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                              "CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                            "CrossAttnUpBlock2D"),
            layers_per_block=2, cross_attention_dim=768, attention_head_dim=8,
            norm_num_groups=32)

# three levels at 32x32 latents: keeps the stored-cross-map inventory that LocalBlend indexes
# (down_cross[2:4] and up_cross[:3] are the 16x16 maps, as in SD-1.x)
TINY = dict(in_channels=4, out_channels=4, sample_size=32,
            block_out_channels=(64, 128, 128),
            down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
            layers_per_block=2, cross_attention_dim=64, attention_head_dim=2,
            norm_num_groups=32)


class UNetOutput(dict):
    @property
    def sample(self):
        return self["sample"]


class PlainProcessor:
    """softmax(scale q k^T) v, ignoring the P2P kwargs (what diffusers' AttnProcessor does)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None,
                 temb=None, use_controller=True, save_attn=True):
        q = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k, v = attn.to_k(ctx), attn.to_v(ctx)
        q, k, v = map(attn.head_to_batch_dim, (q, k, v))
        p = attn.get_attention_scores(q, k)
        o = attn.batch_to_head_dim(torch.bmm(p, v))
        return attn.to_out[1](attn.to_out[0](o))


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        kd = ctx_dim if ctx_dim is not None else dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(kd, dim, bias=False)
        self.to_v = nn.Linear(kd, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = PlainProcessor()

    def prepare_attention_mask(self, mask, seq_len, batch):
        return mask

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(
            b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        return t.reshape(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3).reshape(
            bh // self.heads, n, d * self.heads)

    def get_attention_scores(self, q, k, mask=None):
        s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype),
                          q, k.transpose(-1, -2), beta=0, alpha=self.scale)
        return s.softmax(dim=-1)

    def forward(self, x, encoder_hidden_states=None, **kw):
        return self.processor(self, x, encoder_hidden_states=encoder_hidden_states, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


# Storage-emulation knob for the tolerance analysis (tests/test_gpu_unet.py::test_sd15_unet_forward_full_size): a callable
# applied to every RESIDUAL-STREAM sum (the tensors the HIP path stores between blocks: ResNet output, the three sums of a
# transformer block, proj_out + input).  None = plain fp32, i.e. the oracle proper.
RESID_STORE = None


def _rs(x):
    return x if RESID_STORE is None else RESID_STORE(x)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx, kw):
        x = _rs(self.attn1(self.norm1(x), None, **kw) + x)
        x = _rs(self.attn2(self.norm2(x), ctx, **kw) + x)
        return _rs(self.ff(self.norm3(x)) + x)


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, ctx_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim)])
        self.proj_out = nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx, kw):
        b, c, h, w = x.shape
        res = x
        t = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(b, h * w, c)
        for blk in self.transformer_blocks:
            t = blk(t, ctx, kw)
        t = t.reshape(b, h, w, c).permute(0, 3, 1, 2)
        return _rs(self.proj_out(t) + res)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        # the remaining attribute surface of diffusers' ResnetBlock2D that the reference's Plug-and-Play hook
        # reads when it replaces this forward (text-guided/plug_n_play/pnp_utils.py:96-150)
        self.nonlinearity = F.silu
        self.upsample = self.downsample = None
        self.time_embedding_norm = "default"
        self.dropout = nn.Dropout(0.0)
        self.output_scale_factor = 1.0

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return _rs(x + h)


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Block(nn.Module):
    """Down / up block: resnets (+ attentions) (+ one down/upsampler)."""

    def __init__(self, res_io, attn, heads, ctx_dim, temb_dim, groups, down=None, up=None):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(i, o, temb_dim, groups) for i, o in res_io])
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(o, heads, ctx_dim, groups) for _, o in res_io])
        else:
            self.attentions = None
        if down is not None:
            self.downsamplers = nn.ModuleList([Downsample2D(down)])
        if up is not None:
            self.upsamplers = nn.ModuleList([Upsample2D(up)])


class MidBlock(nn.Module):
    def __init__(self, c, heads, ctx_dim, temb_dim, groups):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, ctx_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_dim, groups) for _ in range(2)])


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


def sinusoid(t, dim):
    """flip_sin_to_cos=True, freq_shift=0: [cos | sin] of t * 10000^(-i/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([a.cos(), a.sin()], dim=-1)


class UNet2DConditionModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        c = dict(SD15)
        c.update(cfg)
        self.cfg = c
        ch = c["block_out_channels"]
        L = c["layers_per_block"]
        heads, ctx, g = c["attention_head_dim"], c["cross_attention_dim"], c["norm_num_groups"]
        self.in_channels = c["in_channels"]
        self.sample_size = c["sample_size"]
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(c["in_channels"], ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, typ in enumerate(c["down_block_types"]):
            inp, out = out, ch[i]
            last = i == len(ch) - 1
            io = [(inp if j == 0 else out, out) for j in range(L)]
            self.down_blocks.append(Block(io, typ.startswith("CrossAttn"), heads, ctx, temb, g,
                                          down=None if last else out))
        self.mid_block = MidBlock(ch[-1], heads, ctx, temb, g)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        out = rev[0]
        for i, typ in enumerate(c["up_block_types"]):
            prev, out = out, rev[i]
            inp = rev[min(i + 1, len(ch) - 1)]
            last = i == len(ch) - 1
            io = []
            for j in range(L + 1):
                skip = inp if j == L else out
                rin = prev if j == 0 else out
                io.append((rin + skip, out))
            self.up_blocks.append(Block(io, typ.startswith("CrossAttn"), heads, ctx, temb, g,
                                        up=None if last else out))
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], c["out_channels"], 3, padding=1)

    # ---- processor registry (diffusers naming)
    def _attn_modules(self):
        out = {}
        for name, m in self.named_modules():
            if isinstance(m, Attention):
                out[name + ".processor"] = m
        return out

    @property
    def attn_processors(self):
        return {k: m.processor for k, m in self._attn_modules().items()}

    def set_attn_processor(self, procs):
        mods = self._attn_modules()
        for k, p in procs.items():
            mods[k].processor = p

    def zero_grad(self, *a, **k):
        return None

    def forward(self, sample, timestep=None, encoder_hidden_states=None,
                cross_attention_kwargs=None):
        kw = dict(cross_attention_kwargs or {})
        b = sample.shape[0]
        t = torch.as_tensor(timestep).reshape(-1)
        if t.numel() == 1:
            t = t.expand(b)
        temb = self.time_embedding(sinusoid(t, self.cfg["block_out_channels"][0]).to(sample.dtype))
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(h, temb)
                if blk.attentions is not None:
                    h = blk.attentions[j](h, encoder_hidden_states, kw)
                skips.append(h)
            if hasattr(blk, "downsamplers"):
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, temb)
        h = self.mid_block.attentions[0](h, encoder_hidden_states, kw)
        h = self.mid_block.resnets[1](h, temb)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(torch.cat([h, skips.pop()], dim=1), temb)
                if blk.attentions is not None:
                    h = blk.attentions[j](h, encoder_hidden_states, kw)
            if hasattr(blk, "upsamplers"):
                h = blk.upsamplers[0](h)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return UNetOutput(sample=h)


def seeded_state_dict(cfg, seed=0):
    """Synthetic weights (no checkpoints exist offline): W ~ N(0, 1/fan_in),
    small random biases and norm affine terms so that bias/affine bugs are visible."""
    net = UNet2DConditionModel(**cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in net.state_dict().items():
        if v.dim() >= 2:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) / math.sqrt(fan_in)
        elif "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif "norm" in k:
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = 0.02 * torch.randn(v.shape, generator=g)
    return sd


def build(cfg=None, seed=0):
    cfg = dict(cfg or SD15)
    net = UNet2DConditionModel(**cfg)
    net.load_state_dict(seeded_state_dict(cfg, seed))
    for p in net.parameters():
        p.requires_grad_(False)
    return net.eval()
