"""The duck-typed ``model`` object the reference's loops receive (a diffusers
StableDiffusionPipeline): ``.unet .scheduler .tokenizer .text_encoder .vae .device``
(text-guided/inversion/p2p_h_edit.py:567-592,613; inversion_utils.py:25-33,50-52)."""
import torch

from .scheduler import DDIMScheduler
from .text import ClipTextEncoder, WordTokenizer
from .unet import SD15_CONFIG, UNet2DConditionModel


class HEditPipeline:
    def __init__(self, unet, scheduler=None, tokenizer=None, text_encoder=None, vae=None, device=None):
        self.unet = unet
        self.device = torch.device(device) if device is not None else unet.device
        self.scheduler = scheduler or DDIMScheduler()
        self.tokenizer = tokenizer or WordTokenizer()
        self.text_encoder = text_encoder
        self.vae = vae

    @classmethod
    def from_random(cls, config=None, seed=0, device="cuda:0", text_layers=12, vae_config=None, with_vae=False):
        """SD-1.x-shaped pipeline with seeded synthetic weights (no checkpoints offline).
        ``with_vae`` adds the image autoencoder (hedit.vae.AutoencoderKL, SD-1.x shape unless
        ``vae_config`` says otherwise)."""
        cfg = dict(SD15_CONFIG)
        cfg.update(config or {})
        unet = UNet2DConditionModel(cfg, device=device)
        unet.init_random(seed)
        dim = cfg["cross_attention_dim"]
        heads = 12 if dim % 12 == 0 else 4
        enc = ClipTextEncoder(dim=dim, layers=text_layers, heads=heads, seed=seed + 7).to(device)
        vae = None
        if with_vae or vae_config is not None:
            from .vae import AutoencoderKL
            vae = AutoencoderKL(vae_config, device=device)
            vae.init_random(seed + 11)
        return cls(unet, DDIMScheduler(), WordTokenizer(), enc, vae, device)

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("the HIP UNet is bound to its creation device; build one pipeline per GPU")
        return self
