"""Pixel-space DDPM UNet of the face-swapping task, CPU fp32 eager (oracle; see oracle/__init__.py).

Restates ``Model`` of the reference's face-swapping/diffusion/diffusion.py:192-341 with its blocks
(:6-24 timestep embedding, :27-75 resampling, :77-138 ResnetBlock, :141-189 AttnBlock).  Unlike the
Stable-Diffusion networks this one is IN the reference tree, so the restatement is PINNED: golden
vectors come from running the reference class itself at toy size (tests/golden/make_golden.py::gen_face,
g11_face.npz), state_dict key names are the reference's.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

CELEBA_HQ = dict(in_channels=3, out_ch=3, ch=128, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2,
                 attn_resolutions=(16,), image_size=256)
TINY_DDPM = dict(in_channels=3, out_ch=3, ch=64, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,), image_size=32)


def timestep_embedding(t, dim):
    """[sin | cos](t * 10000^(-i / (dim/2 - 1)))   (diffusion.py:6-24)"""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    a = t.float()[:, None] * freq[None]
    e = torch.cat([a.sin(), a.cos()], dim=1)
    return F.pad(e, (0, 1)) if dim % 2 else e


def gn(c):
    return nn.GroupNorm(32, c, eps=1e-6)


class Res(nn.Module):
    def __init__(self, cin, cout, temb_ch):
        super().__init__()
        self.norm1, self.conv1 = gn(cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.temb_proj = nn.Linear(temb_ch, cout)
        self.norm2, self.conv2 = gn(cout), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x))) + self.temb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))          # dropout 0.0 in every configuration the reference uses
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + h


class Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = gn(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        n = self.norm(x)
        q = self.q(n).reshape(b, c, h * w).transpose(1, 2)
        k = self.k(n).reshape(b, c, h * w)
        v = self.v(n).reshape(b, c, h * w)
        p = torch.softmax(q @ k * c ** -0.5, dim=2)             # (b, query, key)
        o = (v @ p.transpose(1, 2)).reshape(b, c, h, w)
        return x + self.proj_out(o)


class Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Model(nn.Module):
    def __init__(self, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2,
                 attn_resolutions=(16,), image_size=256, **_unused):
        super().__init__()
        self.ch, self.in_channels, self.resolution = ch, in_channels, image_size
        self.num_res_blocks, L = num_res_blocks, len(ch_mult)
        tc = 4 * ch
        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, tc), nn.Linear(tc, tc)])
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        mult_in = (1,) + tuple(ch_mult)
        res, cin = image_size, ch
        self.down = nn.ModuleList()
        for i in range(L):
            lv = nn.Module()
            lv.block, lv.attn = nn.ModuleList(), nn.ModuleList()
            cin, cout = ch * mult_in[i], ch * ch_mult[i]
            for _ in range(num_res_blocks):
                lv.block.append(Res(cin, cout, tc))
                cin = cout
                if res in attn_resolutions:
                    lv.attn.append(Attn(cin))
            if i != L - 1:
                lv.downsample = Down(cin)
                res //= 2
            self.down.append(lv)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = Res(cin, cin, tc), Attn(cin), Res(cin, cin, tc)
        self.up = nn.ModuleList()
        for i in reversed(range(L)):
            lv = nn.Module()
            lv.block, lv.attn = nn.ModuleList(), nn.ModuleList()
            cout, skip = ch * ch_mult[i], ch * ch_mult[i]
            for j in range(num_res_blocks + 1):
                if j == num_res_blocks:
                    skip = ch * mult_in[i]
                lv.block.append(Res(cin + skip, cout, tc))
                cin = cout
                if res in attn_resolutions:
                    lv.attn.append(Attn(cin))
            if i != 0:
                lv.upsample = Up(cin)
                res *= 2
            self.up.insert(0, lv)
        self.norm_out, self.conv_out = gn(cin), nn.Conv2d(cin, out_ch, 3, padding=1)

    def forward(self, x, t):
        temb = self.temb.dense[1](F.silu(self.temb.dense[0](timestep_embedding(t, self.ch))))
        hs = [self.conv_in(x)]
        for lv in self.down:
            for j, blk in enumerate(lv.block):
                h = blk(hs[-1], temb)
                if len(lv.attn):
                    h = lv.attn[j](h)
                hs.append(h)
            if hasattr(lv, "downsample"):
                hs.append(lv.downsample(hs[-1]))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(hs[-1], temb)), temb)
        for lv in reversed(self.up):
            for j, blk in enumerate(lv.block):
                h = blk(torch.cat([h, hs.pop()], dim=1), temb)
                if len(lv.attn):
                    h = lv.attn[j](h)
            if hasattr(lv, "upsample"):
                h = lv.upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))
