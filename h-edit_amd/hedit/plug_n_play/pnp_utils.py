"""Plug-and-Play feature / attention injection, host side -- mirrors text-guided/plug_n_play/pnp_utils.py:
``get_timesteps`` (:3-10), ``register_time`` (:12-27), ``register_attention_control_efficient`` (:29-93) and
``register_conv_control_efficient`` (:95-154).  The reference patches Python forwards into eight decoder
self-attention modules and one ResNet block; here the three register_* calls only record the schedules and the
current timestep on ``model.unet``, and every UNet call compiles them into the plan the HIP executor reads:

  * self-attention of up_blocks[1].attentions[1:], up_blocks[2], up_blocks[3] (transformer blocks >= 8 in SD-1.x,
    any token count), timestep in the schedule: the rows of the second half... precisely, as the reference writes it
    (:46-55), ONLY when the batch has 2 or 3 rows (``q.shape[0] // 2 == 1``): row 1 uses the q and k of row 0
    -> ``qk_src`` + ``qk_first_block`` + ``qk_max_tokens``;
  * conv2 output of up_blocks[1].resnets[1], timestep in the schedule, same batch condition (:131-140): row 1 takes
    row 0's -> ``feat_src`` + ``feat_resblock`` (the executor copies the conv2 INPUT rows, which is the same thing)."""
import torch

from .. import _lib


def get_timesteps(scheduler, num_inference_steps, strength, device):
    init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
    t_start = max(num_inference_steps - init_timestep, 0)
    return scheduler.timesteps[t_start:], num_inference_steps - t_start


def register_time(model, t):
    model.unet._pnp_t = int(t)


def register_attention_control_efficient(model, injection_schedule):
    model.unet._pnp_qk_schedule = None if injection_schedule is None else [int(v) for v in injection_schedule]
    model.unet._attention_editor = _PnPEditor(model.unet)


def register_conv_control_efficient(model, injection_schedule):
    model.unet._pnp_conv_schedule = None if injection_schedule is None else [int(v) for v in injection_schedule]
    model.unet._attention_editor = _PnPEditor(model.unet)


def _block_indices(cfg):
    """(first transformer block with q/k injection, index of up_blocks[1].resnets[1]) in call order."""
    lpb = cfg["layers_per_block"]
    down, up = cfg["down_block_types"], cfg["up_block_types"]
    if len(up) < 4 or not up[1].startswith("CrossAttn"):
        raise NotImplementedError("Plug-and-Play indexes up_blocks[1..3] of a four-level SD-1.x UNet")
    n_down_attn = sum(lpb for t in down if t.startswith("CrossAttn"))
    n_up0_attn = (lpb + 1) if up[0].startswith("CrossAttn") else 0
    first_tblock = n_down_attn + 1 + n_up0_attn + 1          # up_blocks[1].attentions[1]
    resblock = len(down) * lpb + 2 + (lpb + 1) + 1           # up_blocks[1].resnets[1]
    return first_tblock, resblock


class _PnPEditor:
    """Compiles the recorded schedules into a plan per UNet call (protocol of hedit.engine / hedit.unet)."""

    def __init__(self, unet):
        self.unet = unet
        self.num_att_layers = len(unet.attn_processors)
        self.cur_step = 0
        self._keep = None

    def _active(self, sched):
        t = getattr(self.unet, "_pnp_t", None)
        return sched is not None and t is not None and (t in sched or t == 1000)

    def _plan(self, unet, B, H, W, save_attn, n_images=1):
        """n_images > 1 (lock-step engine): rows [x_orig | src] * n, [x_k | tar] * n -- row n + i takes the q, k / features
        of row i, the reference's two-row rule applied to every image of the batch."""
        if n_images == 1 and B // 2 != 1:  # the reference injects only when q.shape[0] // 2 == 1
            return None
        if n_images > 1 and B != 2 * n_images:
            return None
        qk = self._active(getattr(unet, "_pnp_qk_schedule", None))
        conv = self._active(getattr(unet, "_pnp_conv_schedule", None))
        if not (qk or conv):
            return None
        ar = torch.arange(B, dtype=torch.int32)
        src = ar.clone()
        if n_images > 1:
            src[n_images:] = ar[:n_images]
        else:
            src[1] = 0
        self._keep = (ar.to(unet.device), src.to(unet.device))
        first_tblock, resblock = _block_indices(unet.config)
        p = _lib.P2PPlan()
        p.mode = 1
        p.n_pairs = 0
        p.singles = self._keep[0].data_ptr()
        p.n_single = B
        if qk:
            p.qk_src = self._keep[1].data_ptr()
            p.qk_first_block = first_tblock
            p.qk_max_tokens = 1 << 30
        if conv:
            p.feat_src = self._keep[1].data_ptr()
            p.feat_resblock = resblock
        return p

    def _after_pass(self, save_attn):
        self.cur_step += 1

    def step_callback(self, x_t):
        return x_t
