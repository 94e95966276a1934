"""Face-swapping inversion -- drop-in for the reference's face-swapping/inversion/sde_inversion.py:
``sample_xts_from_x0_sde`` (:4-49) and ``inversion_forward_process_sde`` (:51-158).  Same signatures and
return values; ``model`` is hedit.diffusion.Model (HIP) or any callable ``model(x, t_vector) -> eps`` with
``in_channels`` / ``resolution``.  The per-step arithmetic is a handful of elementwise ops on one
(1,3,S,S) tensor and stays in torch; the cost is the UNet evaluation."""
import torch


def sample_xts_from_x0_sde(model, x0, betas, seq, num_inference_steps=100):
    torch.manual_seed(42)
    torch.cuda.manual_seed(42)
    alpha_bar = (1.0 - betas).cumprod(dim=0)
    sqrt_one_minus_alpha_bar = (1 - alpha_bar) ** 0.5
    t_to_idx = {int(v): k for k, v in enumerate(seq)}
    shape = (num_inference_steps + 1, model.in_channels, model.resolution, model.resolution)
    xts = torch.zeros(shape).to(x0.device)
    noise_added = torch.zeros(shape).to(x0.device)
    xts[0] = x0
    for t in reversed(seq):
        idx = num_inference_steps - t_to_idx[int(t)]
        noise = torch.randn_like(x0)
        xts[idx] = x0 * (alpha_bar[t] ** 0.5) + noise * sqrt_one_minus_alpha_bar[t]
        noise_added[idx] = noise
    return xts, noise_added


def inversion_forward_process_sde(model, x0, betas, seq, etas=1.0, num_inference_steps=100, device=None):
    """-> (xt, zs, xts, noise_added) as the reference; zs[idx] is the z_t with which
    x_{t-1} = mu(x_t) + eta c1 z_t reproduces the independently sampled chain."""
    timesteps = seq
    alpha_bar = (1.0 - betas).cumprod(dim=0)
    if etas is None or (type(etas) in [int, float] and etas == 0):
        raise AssertionError("eta = 0 is not supported by the reference either (sde_inversion.py:124)")
    if type(etas) in [int, float]:
        etas = [etas] * num_inference_steps
    xts, noise_added = sample_xts_from_x0_sde(model, x0, betas, seq, num_inference_steps=num_inference_steps)
    zs = torch.zeros(size=(num_inference_steps, model.in_channels, model.resolution, model.resolution), device=device)
    t_to_idx = {int(v): k for k, v in enumerate(timesteps)}
    xt = x0
    n = x0.size(0)
    for i, t in enumerate(timesteps):
        idx = num_inference_steps - t_to_idx[int(t)] - 1
        t_input = torch.ones(n) * t
        if not getattr(model, "accepts_host_timesteps", False):
            t_input = t_input.to(x0.device)
        xt = xts[idx + 1][None]
        with torch.no_grad():
            eps_t = model(xt, t_input)
        xtm1 = xts[idx][None]
        pred_original_sample = (xt - (1 - alpha_bar[t]) ** 0.5 * eps_t) / alpha_bar[t] ** 0.5
        tm1 = timesteps[i + 1] if i < len(timesteps) - 1 else 0
        eta = 0.5
        c1 = (1 - alpha_bar[tm1]).sqrt() * eta
        c2 = (1 - alpha_bar[tm1]).sqrt() * ((1 - eta ** 2) ** 0.5)
        mu_xt = alpha_bar[tm1].sqrt() * pred_original_sample + c2 * eps_t
        z = (xtm1 - mu_xt) / (etas[idx] * c1)
        zs[idx] = z
        xts[idx] = mu_xt + (etas[idx] * c1) * z
    return xt, zs, xts, noise_added
