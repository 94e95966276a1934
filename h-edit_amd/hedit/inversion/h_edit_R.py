"""h-Edit-R for face swapping -- drop-in for the reference's face-swapping/inversion/h_edit_R.py:7-137.
Same signature, defaults and return value, same sequence of eps-network evaluations (one at x_t, two per
implicit optimisation step at x_{t-1}^k, none of them differentiated: the closures differentiate the
ID / LPIPS losses w.r.t. x through the Tweedie map only).  ``model``: hedit.diffusion.Model (the HIP
executor) or any callable ``model(x, t_vector)``; ``idloss`` / ``lpipsloss``: the caller's torch modules
exposing ``get_cosine_loss(x0)`` / ``get_lpips_loss(x0)`` (arcface/arcface_model.py:40-94), either may be
None."""
import torch


def h_Edit_R(model, lpipsloss, idloss, xT, betas, seq, eta=1.0, zs=None, weight_edit_face=50.0,
             optimization_steps=3, after_skip_steps=100, num_inference_steps=100, soft_face_mask=None,
             per_image=False):
    """per_image (addition, default off = the reference's arithmetic): with n > 1 images in lock-step
    (xT (n,3,S,S), zs (T,n,3,S,S)) the losses are batch MEANS, so each image's gradient carries a factor
    1/n; per_image=True multiplies it back so that every image is edited exactly as it would be alone."""
    if type(eta) in [int, float]:
        etas = [eta] * num_inference_steps
    else:
        etas = eta
    assert len(etas) == num_inference_steps
    timesteps = seq
    xt = xT.unsqueeze(0) if xT.dim() < 4 else xT
    op = list(timesteps[-after_skip_steps:])
    t_to_idx = {int(v): k for k, v in enumerate(timesteps[-after_skip_steps:])}
    alpha_bar = (1.0 - betas).cumprod(dim=0)
    n = xt.size(0)
    gscale = float(n) if per_image else 1.0

    host_t = getattr(model, "accepts_host_timesteps", False)

    def eps_at(x, t):
        t_input = torch.ones(n) * t                 # the reference moves it to the GPU (h_edit_R.py:70); the HIP executor
        with torch.no_grad():                       # reads it on the host, which avoids a device sync per evaluation
            return model(x.detach(), t_input if host_t else t_input.to(x.device))

    for i, t in enumerate(op):
        idx = num_inference_steps - t_to_idx[int(t)] - (num_inference_steps - after_skip_steps + 1)
        z = zs[idx] if zs is not None else None
        # x_{t-1}^base from p(x_{t-1} | x_t), the eta = 0.5 kernel of the reference (h_edit_R.py:69-85)
        eps_t = eps_at(xt, t)
        pred_original_sample = (xt.detach() - (1 - alpha_bar[t]) ** 0.5 * eps_t) / alpha_bar[t] ** 0.5
        tm1 = op[i + 1] if i < len(op) - 1 else 0
        c1 = (1 - alpha_bar[tm1]).sqrt() * 0.5
        c2 = (1 - alpha_bar[tm1]).sqrt() * ((1 - 0.5 ** 2) ** 0.5)
        x_tm1 = alpha_bar[tm1].sqrt() * pred_original_sample + c2 * eps_t + (etas[idx] * c1) * z
        xt_prev_opt = x_tm1.clone().detach().requires_grad_(True)
        if tm1 == 0:
            optimization_steps = 0          # and stays 0, as in the reference (h_edit_R.py:89-90)
        sa, s1 = alpha_bar[tm1] ** 0.5, (1 - alpha_bar[tm1]) ** 0.5
        for _ in range(optimization_steps):
            eps_tm1 = eps_at(xt_prev_opt, tm1)
            with torch.enable_grad():
                rho = alpha_bar[tm1].sqrt() * weight_edit_face
                if idloss:
                    x0_pred = (xt_prev_opt - s1 * eps_tm1) / sa          # Tweedie; eps is a constant here
                    id_loss = idloss.get_cosine_loss(x0_pred)
                    g = torch.autograd.grad(outputs=id_loss, inputs=xt_prev_opt)[0]
                    step = (rho * gscale) * g.detach()
                    if soft_face_mask is not None:
                        step = step * soft_face_mask
                    xt_prev_opt = (xt_prev_opt - step).detach().requires_grad_(True)
                eps_tm1 = eps_at(xt_prev_opt, tm1)                        # recomputed at the updated sample (:114-118)
                if lpipsloss:
                    x0_pred = (xt_prev_opt - s1 * eps_tm1) / sa
                    lpips_loss = lpipsloss.get_lpips_loss(x0_pred)
                    g = torch.autograd.grad(outputs=lpips_loss, inputs=xt_prev_opt)[0]
                    xt_prev_opt = (xt_prev_opt - (rho * gscale) * g.detach()).detach().requires_grad_(True)
        xt = xt_prev_opt.detach().requires_grad_(True)
    return xt
