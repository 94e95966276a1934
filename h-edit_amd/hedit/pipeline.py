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
        return cls(unet, DDIMScheduler(), WordTokenizer(stable_ids=True), enc, vae, device)

    @classmethod
    def from_pretrained(cls, path, device="cuda:0", tokenizer=None, text_encoder=None):
        """Load a LOCAL Stable-Diffusion-1.x checkpoint directory in the diffusers layout (what the
        reference's ``StableDiffusionPipeline.from_pretrained(model_id)`` resolves to,
        text-guided/main_p2p.py:104-106).  The UNet and VAE run on the HIP executors; the CLIP
        text encoder / tokenizer are transformers' (PyTorch-ROCm), unless objects are passed in."""
        import os
        from . import checkpoint as CK
        from .vae import AutoencoderKL
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: not a local checkpoint directory (this build never downloads)")
        # in a multi-rank job only rank 0 reads the weight files; the others receive them by one broadcast per blob
        # (RCCL over xGMI, hedit.dist.state_dict_from_rank0).  The small config.json files are read by every rank.
        from . import dist as HD
        ucfg = CK.read_config(os.path.join(path, "unet"))
        unet = UNet2DConditionModel(CK.unet_config(ucfg), device=device)
        usd = HD.state_dict_from_rank0(lambda: CK.read_component(os.path.join(path, "unet"))[1], unet.param_shapes, device=device,
                                       bf16_names=unet.bf16_exact)
        unet.load_state_dict(usd)
        del usd
        vae = None
        if os.path.isdir(os.path.join(path, "vae")):
            vcfg = CK.read_config(os.path.join(path, "vae"))
            vae = AutoencoderKL(CK.vae_config(vcfg), device=device)
            def read_vae():
                sd = vae.current_names(CK.read_component(os.path.join(path, "vae"))[1])
                fix = {}
                for k, shape in vae.param_shapes.items():       # pre-0.18 checkpoints store 1x1 convs as linears and vice versa
                    w = sd[k]
                    fix[k] = w[:, :, None, None] if (w.dim() == 2 and len(shape) == 4) else (w[:, :, 0, 0] if (w.dim() == 4 and len(shape) == 2) else w)
                return fix
            multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
            vsd = (HD.state_dict_from_rank0(read_vae, vae.param_shapes, device=device) if multi
                   else CK.read_component(os.path.join(path, "vae"))[1])
            vae.load_state_dict(vsd)
            del vsd
        if tokenizer is None:
            from transformers import CLIPTokenizer
            tokenizer = CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"), local_files_only=True)
        if text_encoder is None:
            from transformers import CLIPTextModel
            text_encoder = CLIPTextModel.from_pretrained(os.path.join(path, "text_encoder"), local_files_only=True)
            text_encoder = text_encoder.to(device).eval()
        sch = DDIMScheduler(**CK.scheduler_kwargs(os.path.join(path, "scheduler")))
        return cls(unet, sch, tokenizer, text_encoder, vae, device)

    def save_pretrained(self, path, unet_state_dict, vae_state_dict=None):
        """Write the UNet / VAE weights given as state dicts (the executors keep packed bf16 copies
        only) in the layout from_pretrained reads."""
        import os
        from . import checkpoint as CK
        CK.write_component(os.path.join(path, "unet"), {k: (list(v) if isinstance(v, tuple) else v)
                                                        for k, v in self.unet.config.items()}, unet_state_dict)
        if self.vae is not None and vae_state_dict is not None:
            CK.write_component(os.path.join(path, "vae"), {k: (list(v) if isinstance(v, tuple) else v)
                                                           for k, v in self.vae.config.items()}, vae_state_dict)

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("the HIP UNet is bound to its creation device; build one pipeline per GPU")
        return self
