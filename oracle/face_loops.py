"""Face-swapping sampler (oracle; see oracle/__init__.py): restates
  sample_xts_sde / sde_inversion   face-swapping/inversion/sde_inversion.py:4-49, 51-158
  h_edit_r_face                    face-swapping/inversion/h_edit_R.py:7-137
against the same callables the reference uses: ``model(x, t_vector) -> eps`` with attributes
``in_channels`` / ``resolution``, ``idloss.get_cosine_loss(x0)``, ``lpipsloss.get_lpips_loss(x0)``.
PINNED on vectors from running the reference functions (tests/golden/g11_face.npz).
"""
import torch


def _abar(betas):
    return (1.0 - betas).cumprod(dim=0)


def _step_coeffs(ab, tm1, eta=0.5):
    """c1, c2 of the eta = 0.5 kernel both functions hard-code (sde_inversion.py:140-142, h_edit_R.py:81-83)."""
    s = (1 - ab[tm1]).sqrt()
    return s * eta, s * ((1 - eta ** 2) ** 0.5)


def sample_xts_sde(model, x0, betas, seq, T):
    torch.manual_seed(42)                      # the reference reseeds here (sde_inversion.py:21-22)
    ab = _abar(betas)
    pos = {int(v): k for k, v in enumerate(seq)}
    shape = (T + 1, model.in_channels, model.resolution, model.resolution)
    xts, noise = torch.zeros(shape), torch.zeros(shape)
    xts[0] = x0
    for t in reversed(seq):
        idx = T - pos[int(t)]
        n = torch.randn_like(x0)
        xts[idx] = x0 * ab[t] ** 0.5 + n * (1 - ab[t]) ** 0.5
        noise[idx] = n
    return xts, noise


def sde_inversion(model, x0, betas, seq, etas=1.0, T=100):
    """-> (zs (T,C,H,W), xts (T+1,C,H,W)): z_t such that x_{t-1} = mu(x_t) + eta_t c1 z_t reproduces the
    independently sampled chain (sde_inversion.py:106-155)."""
    ab = _abar(betas)
    if type(etas) in (int, float):
        etas = [etas] * T
    xts, _ = sample_xts_sde(model, x0, betas, seq, T)
    zs = torch.zeros((T, model.in_channels, model.resolution, model.resolution))
    pos = {int(v): k for k, v in enumerate(seq)}
    n = x0.size(0)
    for i, t in enumerate(seq):
        idx = T - pos[int(t)] - 1
        xt = xts[idx + 1][None]
        with torch.no_grad():
            eps = model(xt, torch.ones(n) * t)
        x0_hat = (xt - (1 - ab[t]) ** 0.5 * eps) / ab[t] ** 0.5
        tm1 = seq[i + 1] if i < len(seq) - 1 else 0
        c1, c2 = _step_coeffs(ab, tm1)
        mu = ab[tm1].sqrt() * x0_hat + c2 * eps
        z = (xts[idx][None] - mu) / (etas[idx] * c1)
        zs[idx] = z
        xts[idx] = mu + (etas[idx] * c1) * z
    return zs, xts


def h_edit_r_face(model, lpipsloss, idloss, xT, betas, seq, eta=1.0, zs=None, weight_edit_face=50.0,
                  optimization_steps=3, after_skip_steps=100, num_inference_steps=100, soft_face_mask=None):
    T = num_inference_steps
    etas = [eta] * T if type(eta) in (int, float) else eta
    xt = xT.unsqueeze(0) if xT.dim() < 4 else xT
    op = list(seq[-after_skip_steps:])
    pos = {int(v): k for k, v in enumerate(op)}
    ab = _abar(betas)
    n = xt.size(0)
    K = optimization_steps
    for i, t in enumerate(op):
        idx = T - pos[int(t)] - (T - after_skip_steps + 1)
        z = zs[idx] if zs is not None else None
        with torch.no_grad():
            eps = model(xt, torch.ones(n) * t)
        x0_hat = (xt - (1 - ab[t]) ** 0.5 * eps) / ab[t] ** 0.5
        tm1 = op[i + 1] if i < len(op) - 1 else 0
        c1, c2 = _step_coeffs(ab, tm1)
        x = (ab[tm1].sqrt() * x0_hat + c2 * eps + (etas[idx] * c1) * z).clone().detach().requires_grad_(True)
        if tm1 == 0:
            K = 0                          # sticks for the rest of the run, as in the reference (h_edit_R.py:89-90)
        for _ in range(K):
            rho = ab[tm1].sqrt() * weight_edit_face
            for which, loss_of in (("id", idloss.get_cosine_loss if idloss else None),
                                   ("lpips", lpipsloss.get_lpips_loss if lpipsloss else None)):
                if loss_of is None:
                    continue
                with torch.no_grad():
                    eps1 = model(x, torch.ones(n) * tm1)
                with torch.enable_grad():
                    x0p = (x - (1 - ab[tm1]) ** 0.5 * eps1) / ab[tm1] ** 0.5      # eps is a constant here
                    g = torch.autograd.grad(loss_of(x0p), x)[0]
                    step = rho * g.detach()
                    if soft_face_mask is not None and which == "id":       # the mask gates the ID step only (:109-112)
                        step = step * soft_face_mask
                    x = x - step
        xt = x.detach().requires_grad_(True)
    return xt
