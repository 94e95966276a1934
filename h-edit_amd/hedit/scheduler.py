"""DDIM scheduler fields the h-Edit path reads.

The reference uses diffusers' DDIMScheduler (text-guided/main_p2p.py:139-146) but touches only
``alphas``, ``alphas_cumprod``, ``final_alpha_cumprod``, ``timesteps``, ``num_inference_steps``,
``config.num_train_timesteps`` / ``config.timestep_spacing`` and ``set_timesteps``
(text-guided/inversion/inversion_utils.py:50-52,84-87,183; p2p_h_edit.py:572-575,590;
ddpm_inversion.py:22-34).  Defaults are SD-1.4's scheduler config: scaled-linear betas
0.00085..0.012, 1000 train steps, set_alpha_to_one=False, steps_offset=1, "leading" spacing.
"""
import types

import numpy as np
import torch


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
                 steps_offset=1, timestep_spacing="leading"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                   dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps,
                                            steps_offset=steps_offset,
                                            timestep_spacing=timestep_spacing,
                                            beta_start=beta_start, beta_end=beta_end,
                                            beta_schedule=beta_schedule, clip_sample=clip_sample,
                                            set_alpha_to_one=set_alpha_to_one)
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing != "leading":
            raise NotImplementedError("only 'leading' spacing is used by the h-Edit drivers")
        ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)
