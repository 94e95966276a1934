"""CPU checks of the autoencoder restatement (oracle/sd_vae.py): parameter inventory of the
published SD-1.x AutoencoderKL and the surface the reference calls."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd_vae  # noqa: E402


def test_sd15_inventory():
    m = sd_vae.AutoencoderKL()
    sd = m.state_dict()
    assert len(sd) == 248
    assert sum(v.numel() for v in sd.values()) == 83653863
    assert sd["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"].shape == (256, 512, 1, 1)
    assert sd["encoder.down_blocks.0.downsamplers.0.conv.weight"].shape == (128, 128, 3, 3)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert sd["encoder.mid_block.attentions.0.to_out.0.weight"].shape == (512, 512)


def test_surface_and_shapes():
    torch.manual_seed(0)
    m = sd_vae.AutoencoderKL(sd_vae.TINY_VAE).eval()
    x = torch.rand(2, 3, 32, 16) * 2 - 1
    with torch.no_grad():
        z = m.encode(x).latent_dist.mode()
        y = m.decode(z)
    assert z.shape == (2, 4, 16, 8)
    assert y.sample.shape == y["sample"].shape == (2, 3, 32, 16)


def test_encoder_downsample_is_right_bottom_padded():
    """Downsample2D(padding=0): zero pad (0,1,0,1) then stride-2 valid conv -- output (oy,ox) reads
    input rows 2oy..2oy+2; the last tap row/column falls on the zero pad."""
    torch.manual_seed(1)
    blk = sd_vae.DownBlock(32, 32, 1, 32, True).eval()
    x = torch.randn(1, 32, 8, 8)
    with torch.no_grad():
        r = blk.resnets[0](x)
        want = F.conv2d(F.pad(r, (0, 1, 0, 1)), blk.downsamplers[0].conv.weight, blk.downsamplers[0].conv.bias, stride=2)
        got = blk(x)
    assert got.shape == (1, 32, 4, 4)
    assert torch.allclose(got, want)
