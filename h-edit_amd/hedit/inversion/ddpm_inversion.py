"""Edit-friendly DDPM inversion (mirrors text-guided/inversion/ddpm_inversion.py:
sample_xts_from_x0 :5-52, inversion_forward_process_ddpm :54-167) on the HIP UNet."""
import torch

from ..engine import HEditEngine


def sample_xts_from_x0(model, x0, num_inference_steps=50):
    """Independent draws x_t ~ q(x_t | x_0) for every scheduler timestep (reference
    ddpm_inversion.py:5-52): returns (xts, noise_added), both (T+1, C, H, W) with row 0 = x0 / zeros
    and row idx = T - position of t in ``scheduler.timesteps``."""
    ab = model.scheduler.alphas_cumprod
    ts = [int(t) for t in model.scheduler.timesteps]
    x = x0 if x0.dim() == 3 else x0[0]
    xts = torch.zeros((num_inference_steps + 1,) + tuple(x.shape), device=x0.device)
    noise_added = torch.zeros_like(xts)
    xts[0] = x
    for pos in reversed(range(len(ts))):
        t = ts[pos]
        idx = num_inference_steps - pos
        nz = torch.randn_like(x)
        xts[idx] = x * float(ab[t]) ** 0.5 + nz * float(1 - ab[t]) ** 0.5
        noise_added[idx] = nz
    return xts, noise_added


def inversion_forward_process_ddpm(model, x0, etas=None, prog_bar=True, prompt="", cfg_scale_src=1.0,
                                   cfg_scale_src_edit=3.5, num_inference_steps=50, noise=None, generator=None):
    """Returns (xt, zs, xts, noise_added) like the reference: zs (T,C,H,W), xts (T+1,C,H,W).
    `noise` (T+1,C,H,W) lets the caller fix the forward-process noise (parity tests)."""
    if etas is None or (type(etas) in [int, float] and etas == 0):
        raise AssertionError("eta must be > 0 for DDPM inversion")   # reference: assert not eta_is_zero
    if type(etas) in [int, float]:
        eta = float(etas)
    else:
        assert len(etas) == num_inference_steps
        eta = [float(e) for e in etas]        # etas[idx], as the reference indexes them (ddpm_inversion.py:152-161)
    assert model.scheduler.num_inference_steps == num_inference_steps
    eng = HEditEngine(model)
    x = x0 if x0.dim() == 4 else x0[None]
    nz = None if noise is None else noise[:, None]
    zs, xts, noise_added = eng.ddpm_inversion(x, [prompt], eta=eta, cfg_src=cfg_scale_src, noise=nz, generator=generator,
                                              return_noise=True)
    return xts[1], zs[:, 0], xts[:, 0], noise_added[:, 0]
