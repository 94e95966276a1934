"""Build the HIP pipeline and the oracle model on IDENTICAL synthetic weights / text encoder."""
import types

import torch

from hedit.pipeline import HEditPipeline
from hedit.scheduler import DDIMScheduler
from hedit.text import ClipTextEncoder, WordTokenizer
from hedit.unet import SD15_CONFIG, TINY_CONFIG, UNet2DConditionModel, random_state_dict


def make_oracle(config, num_steps, seed=0, text_layers=2, out_scale=1.0):
    """The oracle half of make_pair alone -- no HIP library, no GPU (tests/golden/make_loop_trajectory.py runs it in the build
    container).  The parameter names / shapes are the oracle module's own (the HIP executor reports the same ones:
    make_pair loads one state_dict into both, strictly), and random_state_dict seeds every tensor by name."""
    from oracle import sd_unet as OU
    cfg = dict(config)
    onet = OU.UNet2DConditionModel(**cfg)
    sd = random_state_dict({k: tuple(v.shape) for k, v in onet.state_dict().items()}, seed)
    if out_scale != 1.0:
        sd["conv_out.weight"] = sd["conv_out.weight"] * out_scale
        sd["conv_out.bias"] = sd["conv_out.bias"] * out_scale
    onet.load_state_dict(sd)
    for p in onet.parameters():
        p.requires_grad_(False)
    onet.eval()
    om = types.SimpleNamespace()
    om.device = torch.device("cpu")
    om.unet = onet
    om.scheduler = DDIMScheduler()
    om.scheduler.set_timesteps(num_steps)
    om.tokenizer = WordTokenizer()
    om.text_encoder = ClipTextEncoder(dim=cfg["cross_attention_dim"], layers=text_layers, heads=4, seed=seed + 7)
    om.vae = None
    return om, sd


def make_hip(config, num_steps, seed=0, device="cuda:0", text_layers=2, out_scale=1.0):
    """The HIP half of make_pair alone (same seeded weights / text encoder): for tests that compare with a committed oracle fixture
    instead of running the oracle (tests/test_gpu_loop_trajectory.py)."""
    cfg = dict(config)
    unet = UNet2DConditionModel(cfg, device=device)
    sd = random_state_dict(unet.param_shapes, seed)
    if out_scale != 1.0:
        sd["conv_out.weight"] = sd["conv_out.weight"] * out_scale
        sd["conv_out.bias"] = sd["conv_out.bias"] * out_scale
    unet.load_state_dict(sd)
    enc = ClipTextEncoder(dim=cfg["cross_attention_dim"], layers=text_layers, heads=4, seed=seed + 7)
    hip = HEditPipeline(unet, DDIMScheduler(), WordTokenizer(), enc.to(device), None, device)
    hip.scheduler.set_timesteps(num_steps)
    return hip


def make_pair(config, num_steps, seed=0, device="cuda:0", text_layers=2, out_scale=1.0):
    """returns (hip_model, oracle_model, state_dict).  out_scale damps the synthetic network's
    output layer: a random-weight eps-network at full gain makes the sampler chain chaotic (any
    perturbation grows ~2x per step), which says nothing about kernel correctness."""
    from oracle import sd_unet as OU
    cfg = dict(config)
    unet = UNet2DConditionModel(cfg, device=device)
    sd = random_state_dict(unet.param_shapes, seed)
    if out_scale != 1.0:
        sd["conv_out.weight"] = sd["conv_out.weight"] * out_scale
        sd["conv_out.bias"] = sd["conv_out.bias"] * out_scale
    unet.load_state_dict(sd)
    tok = WordTokenizer()
    dim = cfg["cross_attention_dim"]
    enc = ClipTextEncoder(dim=dim, layers=text_layers, heads=4, seed=seed + 7)
    hip = HEditPipeline(unet, DDIMScheduler(), tok, enc.to(device), None, device)
    hip.scheduler.set_timesteps(num_steps)

    onet = OU.UNet2DConditionModel(**cfg)
    onet.load_state_dict(sd)
    for p in onet.parameters():
        p.requires_grad_(False)
    onet.eval()
    om = types.SimpleNamespace()
    om.device = torch.device("cpu")
    om.unet = onet
    om.scheduler = DDIMScheduler()
    om.scheduler.set_timesteps(num_steps)
    om.tokenizer = tok
    cpu_enc = ClipTextEncoder(dim=dim, layers=text_layers, heads=4, seed=seed + 7)
    om.text_encoder = cpu_enc
    om.vae = None
    return hip, om, sd
