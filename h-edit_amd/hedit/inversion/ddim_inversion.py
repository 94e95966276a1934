"""DDIM inversion for h-Edit-D (mirrors text-guided/inversion/ddim_inversion.py: next_step :7-28,
get_noise_pred :30-52, ddim_inversion :54-131) on the HIP UNet."""
from ..engine import HEditEngine


def ddim_inversion(model, w0, prompt, cfg_scale):
    """Returns (latent, zs, latents) like the reference: zs (T,C,H,W); latents = list of T+1
    tensors (1,C,H,W), so ``xT = latents[after_skip_steps]`` works as in main_p2p.py:218-236."""
    x = w0 if w0.dim() == 4 else w0[None]
    lat, zs, lats = HEditEngine(model).ddim_inversion(x, [prompt], cfg_scale)
    return lat, zs[:, 0], [lats[i] for i in range(lats.shape[0])]
