"""SD-1.x image autoencoder, CPU fp32 eager (oracle; see oracle/__init__.py).

The reference obtains this network from diffusers==0.18.0 (``model.vae`` of the
StableDiffusionPipeline: ``vae.encode(image).latent_dist.mode()`` at text-guided/main_p2p.py:159
and p2p/ptp_classes.py:351-373, ``vae.decode(1 / 0.18215 * latents).sample`` at main_p2p.py:263).
diffusers is not under /root/reference and not in this image, and the reference holds no golden
vector for it, so this is a restatement of the PUBLISHED architecture (AutoencoderKL /
Encoder / Decoder / UNetMidBlock2D with one single-head attention, resnet eps 1e-6, encoder
Downsample2D(padding=0) = F.pad (0,1,0,1) + stride-2 conv, decoder nearest-2x + conv)
=> PARITY UNPINNED for the body, like oracle/sd_unet.py.  state_dict keys are diffusers' names.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15_VAE = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
# two levels (f = 2), 64/128 channels: small enough for CPU parity runs
TINY_VAE = dict(in_channels=3, latent_channels=4, block_out_channels=(64, 128),
                layers_per_block=1, norm_num_groups=32, scaling_factor=0.18215)


class Res(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if hasattr(self, "conv_shortcut"):
            x = self.conv_shortcut(x)
        return x + h


class Attn(nn.Module):
    """diffusers Attention(heads=1, residual_connection=True, norm_num_groups=32, bias=True)."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(c), dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(b, c, h, w)


class Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([Res(c, c, groups), Res(c, c, groups)])
        self.attentions = nn.ModuleList([Attn(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Conv(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=0 if stride == 2 else 1)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([Res(cin if j == 0 else cout, cout, groups) for j in range(n)])
        if down:
            self.downsamplers = nn.ModuleList([_Conv(cout, 2)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))
        return x


class UpBlock(nn.Module):
    def __init__(self, cin, cout, n, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([Res(cin if j == 0 else cout, cout, groups) for j in range(n)])
        if up:
            self.upsamplers = nn.ModuleList([_Conv(cout, 1)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, n = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["layers_per_block"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([DownBlock(ch[max(i - 1, 0)], ch[i], n, g, i < len(ch) - 1) for i in range(len(ch))])
        self.mid_block = Mid(ch[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, n = list(reversed(cfg["block_out_channels"])), cfg["norm_num_groups"], cfg["layers_per_block"]
        self.conv_in = nn.Conv2d(cfg["latent_channels"], ch[0], 3, padding=1)
        self.mid_block = Mid(ch[0], g)
        self.up_blocks = nn.ModuleList([UpBlock(ch[max(i - 1, 0)], ch[i], n + 1, g, i < len(ch) - 1) for i in range(len(ch))])
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], cfg["in_channels"], 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class _Out(dict):
    @property
    def sample(self):
        return self["sample"]


class _Dist:
    def __init__(self, moments):
        self.mean, self.logvar = moments.chunk(2, dim=1)

    def mode(self):
        return self.mean


class _EncOut:
    def __init__(self, moments):
        self.latent_dist = _Dist(moments)


class AutoencoderKL(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        cfg = dict(SD15_VAE)
        cfg.update(config or {})
        self.config = cfg
        self.encoder, self.decoder = Encoder(cfg), Decoder(cfg)
        lc = cfg["latent_channels"]
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)

    def encode(self, x):
        return _EncOut(self.quant_conv(self.encoder(x)))

    def decode(self, z):
        return _Out(sample=self.decoder(self.post_quant_conv(z)))
