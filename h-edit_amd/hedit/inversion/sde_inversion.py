"""Face-swapping inversion -- drop-in for the two functions of the reference's face-swapping/inversion/sde_inversion.py:
``sample_xts_from_x0_sde`` (:4-49) draws x_t ~ q(x_t | x_0) independently for every kept timestep (reseeding with 42,
as the reference does), ``inversion_forward_process_sde`` (:51-158) then walks the timesteps downwards and solves, per
step, for the noise z_t that makes the reverse kernel (the eta = 0.5 variant the reference hard-codes) reproduce the
sampled x_{t-1}; it writes that exact x_{t-1} back so no error accumulates.  Same signatures and return values.
``model``: hedit.diffusion.Model (HIP) or any callable ``model(x, t_vector) -> eps`` exposing ``in_channels`` /
``resolution``.  The per-step arithmetic is a handful of elementwise ops on one image; the cost is the eps evaluation."""
import torch

_KERNEL_ETA = 0.5        # fixed inside the reference functions (sde_inversion.py:140, h_edit_R.py:81)


def _alpha_bar(betas):
    return torch.cumprod(1.0 - betas, dim=0)


def _slot(seq, num_inference_steps):
    """timestep -> row of xts / noise (1 for the smallest kept timestep ... num_inference_steps for the largest)"""
    return {int(t): num_inference_steps - k for k, t in enumerate(seq)}


def sample_xts_from_x0_sde(model, x0, betas, seq, num_inference_steps=100):
    torch.manual_seed(42)
    torch.cuda.manual_seed(42)
    ab = _alpha_bar(betas)
    rows = (num_inference_steps + 1, model.in_channels, model.resolution, model.resolution)
    xts = torch.zeros(rows, device=x0.device)
    noise_added = torch.zeros(rows, device=x0.device)
    xts[0] = x0
    slot = _slot(seq, num_inference_steps)
    for t in reversed(seq):                       # ascending timesteps: the order fixes the random stream
        eps = torch.randn_like(x0)
        xts[slot[int(t)]] = ab[t].sqrt() * x0 + (1 - ab[t]).sqrt() * eps
        noise_added[slot[int(t)]] = eps
    return xts, noise_added


def inversion_forward_process_sde(model, x0, betas, seq, etas=1.0, num_inference_steps=100, device=None):
    """-> (xt, zs, xts, noise_added); zs[i] takes x at row i + 1 of xts to row i."""
    if etas is None or (isinstance(etas, (int, float)) and etas == 0):
        raise AssertionError("eta = 0 is not supported by the reference either (sde_inversion.py:124)")
    if isinstance(etas, (int, float)):
        etas = [etas] * num_inference_steps
    ab = _alpha_bar(betas)
    xts, noise_added = sample_xts_from_x0_sde(model, x0, betas, seq, num_inference_steps=num_inference_steps)
    zs = torch.zeros((num_inference_steps,) + tuple(xts.shape[1:]), device=device)
    host_t = getattr(model, "accepts_host_timesteps", False)
    n = x0.size(0)
    slot = _slot(seq, num_inference_steps)
    xt = x0
    for i, t in enumerate(seq):
        row = slot[int(t)] - 1                    # this step produces xts[row] from xts[row + 1]
        t_vec = torch.ones(n) * t
        xt = xts[row + 1][None]
        with torch.no_grad():
            eps = model(xt, t_vec if host_t else t_vec.to(x0.device))
        t_next = seq[i + 1] if i + 1 < len(seq) else 0
        x0_hat = (xt - (1 - ab[t]).sqrt() * eps) / ab[t].sqrt()
        spread = (1 - ab[t_next]).sqrt()
        mean = ab[t_next].sqrt() * x0_hat + spread * (1 - _KERNEL_ETA ** 2) ** 0.5 * eps
        sigma = etas[row] * spread * _KERNEL_ETA
        zs[row] = (xts[row][None] - mean) / sigma
        xts[row] = mean + sigma * zs[row]
    return xt, zs, xts, noise_added
