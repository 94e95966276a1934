"""Reading a local Stable-Diffusion checkpoint directory in the diffusers layout

    <dir>/unet/{config.json, diffusion_pytorch_model.safetensors | .bin}
    <dir>/vae/{config.json, diffusion_pytorch_model.safetensors | .bin}
    <dir>/scheduler/scheduler_config.json
    <dir>/text_encoder/, <dir>/tokenizer/        (transformers CLIPTextModel / CLIPTokenizer)

which is what ``StableDiffusionPipeline.from_pretrained(model_id)`` resolves to in the reference
(text-guided/main_p2p.py:104-106; "model_id = stable_diff_local" for local copies).  Only local
directories: this build never touches the network.
"""
import json
import os

import torch

_WEIGHT_FILES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin",
                 "diffusion_pytorch_model.fp16.safetensors")


def read_config(path):
    with open(os.path.join(path, "config.json")) as f:
        return json.load(f)


def read_component(path):
    """(config dict, state_dict of CPU tensors) of one sub-folder."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    for name in _WEIGHT_FILES:
        p = os.path.join(path, name)
        if not os.path.exists(p):
            continue
        if p.endswith(".safetensors"):
            from safetensors.torch import load_file
            return cfg, load_file(p, device="cpu")
        return cfg, torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no {' / '.join(_WEIGHT_FILES)} under {path}")


def write_component(path, config, state_dict):
    """inverse of read_component (safetensors); used to export synthetic weights and by the tests."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(config, f, indent=1)
    save_file({k: v.detach().cpu().contiguous() for k, v in state_dict.items()},
              os.path.join(path, "diffusion_pytorch_model.safetensors"))


_UNET_KEYS = ("in_channels", "out_channels", "sample_size", "block_out_channels", "down_block_types",
              "up_block_types", "layers_per_block", "cross_attention_dim", "attention_head_dim", "norm_num_groups")
_VAE_KEYS = ("in_channels", "latent_channels", "block_out_channels", "layers_per_block", "norm_num_groups",
             "scaling_factor")


def unet_config(cfg):
    """the fields of diffusers' UNet2DConditionModel config this build reads; anything that would
    change the architecture away from SD-1.x is rejected instead of being ignored"""
    out = {k: cfg[k] for k in _UNET_KEYS if k in cfg}
    if isinstance(out.get("attention_head_dim"), (list, tuple)):
        vals = set(out["attention_head_dim"])
        if len(vals) != 1:
            raise NotImplementedError("per-level attention_head_dim (SD-2.x style) is not supported")
        out["attention_head_dim"] = vals.pop()
    for k in ("use_linear_projection", "dual_cross_attention", "only_cross_attention", "upcast_attention",
              "class_embed_type", "addition_embed_type", "time_embedding_type_override"):
        v = cfg.get(k)
        if isinstance(v, (list, tuple)):
            v = any(v)
        if v:      # anything but False / None / absent
            raise NotImplementedError(f"UNet config {k}={cfg[k]!r} is outside the SD-1.x architecture this build implements")
    for k in ("block_out_channels", "down_block_types", "up_block_types"):
        if k in out:
            out[k] = tuple(out[k])
    return out


def vae_config(cfg):
    out = {k: cfg[k] for k in _VAE_KEYS if k in cfg}
    if "block_out_channels" in out:
        out["block_out_channels"] = tuple(out["block_out_channels"])
    return out


def scheduler_kwargs(path):
    p = os.path.join(path, "scheduler_config.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        cfg = json.load(f)
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "clip_sample", "set_alpha_to_one",
            "steps_offset", "timestep_spacing")
    return {k: cfg[k] for k in keys if k in cfg}
