"""h-Edit sampling loops + DDPM inversion (oracle; see oracle/__init__.py).

Restates, against the same duck-typed ``model`` object the reference uses:
  encode_text                   text-guided/inversion/inversion_utils.py:13-35
  sample_xts / ddpm_inversion   text-guided/inversion/ddpm_inversion.py:5-52, 54-167
  ddim_inversion                text-guided/inversion/ddim_inversion.py:7-131
  h_edit_r_explicit             text-guided/inversion/p2p_h_edit.py:21-156
  h_edit_r_implicit             text-guided/inversion/p2p_h_edit.py:162-362
  h_edit_p2p_explicit           text-guided/inversion/p2p_h_edit.py:380-523
  h_edit_p2p_implicit           text-guided/inversion/p2p_h_edit.py:529-701
  h_edit_p2p_implicit_style     text-guided-n-style/inversion/h_edit.py:14-192 (text + style editing)
  h_edit_masactrl_implicit      text-guided/inversion/masactrl_h_edit.py:10-155 (editor from oracle/masactrl.py)
  h_edit_pnp_implicit           text-guided/inversion/pnp_h_edit.py:24-167 (hooks from oracle/pnp.py)
The four loops share one skeleton here (``_loop``); the per-variant differences are the UNet
batches and which eps feeds the three CFG mixes.
"""
import torch

from . import sched as S


def encode_text(model, prompts):
    ids = model.tokenizer(prompts, padding="max_length",
                          max_length=model.tokenizer.model_max_length, truncation=True,
                          return_tensors="pt").input_ids
    with torch.no_grad():
        return model.text_encoder(ids.to(model.device))[0]


# --------------------------------------------------------------------------- inversion
def sample_xts(model, x0, T):
    """x_t ~ q(x_t | x_0) independently for every step (ddpm_inversion.py:22-52).  Draws one
    ``randn_like(x0)`` per timestep, walking the timesteps from small to large."""
    sch = model.scheduler
    ab = sch.alphas_cumprod
    ts = sch.timesteps
    pos = {int(v): k for k, v in enumerate(ts)}
    c, s = model.unet.in_channels, model.unet.sample_size
    xts = torch.zeros(T + 1, c, s, s)
    noise = torch.zeros(T + 1, c, s, s)
    xts[0] = x0
    for t in reversed(ts):
        i = T - pos[int(t)]
        n = torch.randn_like(x0)
        xts[i] = x0 * ab[t] ** 0.5 + n * (1 - ab[t]) ** 0.5
        noise[i] = n
    return xts, noise


def ddpm_inversion(model, x0, eta=1.0, prompt="", cfg_src=1.0, T=50):
    """Edit-friendly DDPM inversion: z_t such that the eta-chain reproduces the sampled x_t's
    (ddpm_inversion.py:86-162).  Returns (zs[T], xts[T+1], noise)."""
    sch = model.scheduler
    ab = sch.alphas_cumprod
    ts = sch.timesteps
    unc = encode_text(model, "")
    cond = encode_text(model, prompt) if prompt != "" else None
    xts, noise = sample_xts(model, x0, T)
    c, s = model.unet.in_channels, model.unet.sample_size
    zs = torch.zeros(T, c, s, s)
    pos = {int(v): k for k, v in enumerate(ts)}
    for t in ts:
        i = T - pos[int(t)] - 1
        xt = xts[i + 1][None]
        with torch.no_grad():
            e = model.unet.forward(xt, timestep=t, encoder_hidden_states=unc).sample
            if cond is not None:
                ec = model.unet.forward(xt, timestep=t, encoder_hidden_states=cond).sample
                e = e + cfg_src * (ec - e)
        x0_hat = (xt - (1 - ab[t]) ** 0.5 * e) / ab[t] ** 0.5
        a_p = S._abar_prev(sch, t)
        var = S.get_variance(sch, t)
        mu = a_p ** 0.5 * x0_hat + (1 - a_p - (eta ** 2) * var) ** 0.5 * e
        z = (xts[i][None] - mu) / (eta * var ** 0.5)
        zs[i] = z
        xts[i] = mu + (eta * var ** 0.5) * z                      # re-anchor (:160-162)
    return zs, xts, noise


def ddim_inversion(model, w0, prompt, cfg_scale):
    """Deterministic DDIM inversion + the per-step corrections u_t that make the eta=1 chain of
    h-Edit-D reproduce it (text-guided/inversion/ddim_inversion.py: next_step :7-28,
    get_noise_pred :30-52, ddim_inversion :54-131).  Returns (latent_T, zs[T], latents[T+1])."""
    sch = model.scheduler
    ab = sch.alphas_cumprod
    T = sch.num_inference_steps
    step = sch.config.num_train_timesteps // T
    ctx = torch.cat([encode_text(model, ""), encode_text(model, prompt)])

    def eps(x, t):
        with torch.no_grad():
            e = model.unet(torch.cat([x] * 2), t, encoder_hidden_states=ctx)["sample"]
        e_u, e_c = e.chunk(2)
        return e_u + cfg_scale * (e_c - e_u)

    lat = w0.clone().detach()
    lats = [lat]
    for i in range(T):
        t = sch.timesteps[len(sch.timesteps) - i - 1]
        e = eps(lat, t)
        cur = min(t - step, 999)
        a_cur = ab[cur] if cur >= 0 else sch.final_alpha_cumprod
        a_next = ab[t]
        x0 = (lat - (1 - a_cur) ** 0.5 * e) / a_cur ** 0.5
        lat = a_next ** 0.5 * x0 + (1 - a_next) ** 0.5 * e
        lats.append(lat)
    c, sz = model.unet.in_channels, model.unet.sample_size
    zs = torch.zeros(T, c, sz, sz)
    pos = {int(v): k for k, v in enumerate(sch.timesteps)}
    for t in sch.timesteps:
        idx = T - pos[int(t)] - 1
        xt = lats[idx + 1]
        e = eps(xt, t)
        x0 = (xt - (1 - ab[t]) ** 0.5 * e) / (ab[t] ** 0.5)
        a_p = S._abar_prev(sch, t)
        mu = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e
        z = lats[idx] - mu
        zs[idx] = z
        lats[idx] = mu + z
    return lat, zs, lats


# --------------------------------------------------------------------------- loops
def _prep(model, xT, prompts, cfg_scales, after_skip_steps):
    sch = model.scheduler
    assert len(prompts) >= 2, "only support prompt editing"
    cfg = torch.Tensor(cfg_scales).view(-1, 1, 1, 1)
    txt = encode_text(model, prompts)
    unc = encode_text(model, [""] * len(prompts))
    ts = sch.timesteps
    xt = xT.unsqueeze(0) if xT.dim() < 4 else xT
    xt = torch.cat([xt] * len(prompts))
    op = list(ts[-after_skip_steps:])
    return sch, cfg.chunk(3), txt, unc, xt, op


def _rms(v):
    return (v * v).mean().sqrt().item()


def _l1_pull(x, anchor, corr, w):
    """k>0 reconstruction pull (p2p_h_edit.py:670-684): x - rho * d|x-anchor|_1/dx with
    rho = rms(corr)/(rms(grad)+1e-8) * w.  The L1 mean gradient is sign(x-anchor)/numel."""
    xg = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        loss = torch.nn.functional.l1_loss(xg, anchor)
        g = torch.autograd.grad(loss, xg)[0]
    rho = _rms(corr) / (_rms(g) + 1e-8) * w
    return x - rho * g


def _style_step(model, image_encoder, x, e_tar, corr, tt, weight):
    """Style-guidance update of x_{t-1}^k (n-style h_edit.py:162-185): Tweedie x0 from the target
    eps at t-1 -> vae.decode(x0 / 0.18215) -> ||Gram residual||_F of the image encoder -> gradient
    w.r.t. x through decoder and encoder -> x - rho g, rho = rms(correction) / rms(g) * weight."""
    xs = x.clone().detach().requires_grad_(True)
    with torch.enable_grad():
        x0 = S.tweedie_x0(model.scheduler, e_tar, tt, xs)
        img = model.vae.decode(1 / 0.18215 * x0).sample
        loss = torch.linalg.norm(image_encoder.get_gram_matrix_residual(img))
        g = torch.autograd.grad(outputs=loss, inputs=xs)[0]
    rho = _rms(corr) / _rms(g) * weight
    return (xs - rho * g.detach()).detach()


def _loop(model, xT, eta, prompts, cfg_scales, zs, controller, after_skip_steps, ddim_inv,
          p2p, implicit, K=1, w_rec=0.1, style=None, masactrl=False):
    """style = (image_encoder | None, weight_edit_clip) selects the n-style variant of the implicit
    P2P loop: no reconstruction pull between inner steps (rec_term = x^k, h_edit.py:149) and one
    style-guidance update after every text update (h_edit.py:160-188)."""
    sch, (w_src, w_hat, w_tar), txt, unc, xt, op = _prep(model, xT, prompts, cfg_scales,
                                                          after_skip_steps)
    T = sch.num_inference_steps
    if not p2p:
        assert not ddim_inv, "only support prompt editing and DDPM sampling"
    # masactrl: same skeleton as the implicit P2P loop; the registered attention editor is switched off with
    # `use_editor`, the edit pass carries no kwargs, no reconstruction pull, no LocalBlend callback
    off = {"use_editor": False} if masactrl else {"use_controller": False}
    pos = {int(v): k for k, v in enumerate(op)}

    def unet(x, t, ctx, kw):
        with torch.no_grad():
            return model.unet(x, t, encoder_hidden_states=ctx, cross_attention_kwargs=kw).sample

    def mixes(e_u_src, e_c_src, e_u_tar, e_c_tar):
        e_hat = e_u_src + w_hat * (e_c_src - e_u_src)
        e_tar = e_u_tar + w_tar * (e_c_tar - e_u_tar)
        return e_tar - e_hat

    ahead = sch.timesteps[-(after_skip_steps + 1)] if after_skip_steps != T else -1

    for i, t in enumerate(op):
        idx = T - pos[int(t)] - (T - after_skip_steps + 1)
        z = zs[idx] if zs is not None else None
        tt = op[i + 1] if i < len(op) - 1 else 0

        # (R-implicit only) one extra correction of the start sample when steps were skipped
        # (p2p_h_edit.py:239-267)
        if (not p2p) and implicit and i == 0 and ahead != -1:
            e = unet(torch.cat([xt[1:]] * 4), t, torch.cat([unc, txt]), off)
            corr = mixes(e[0:1], e[2:3], e[1:2], e[3:4])
            xt[1] = xt[1] + S.edit_coeff(sch, ahead, t, eta, ddim_inv) * corr[0]

        # base pass: eps with the source prompt, CFG weight w_src  -> x_{t-1}^{orig}, x_{t-1}^{base}
        if p2p:
            e = unet(torch.cat([xt] * 2), t, torch.cat([unc[:1], unc[:1], txt[:1], txt[:1]]), off)
        else:
            e = unet(torch.cat([xt[1:]] * 2), t, torch.cat([unc[:1], txt[:1]]), off)
        e_u, e_c = e.chunk(2)
        prev = S.reverse_step(sch, e_u + w_src * (e_c - e_u), t, xt, eta=eta, z=z, ddim_inv=ddim_inv)
        x_orig, x_base = prev.chunk(2)
        coeff = S.edit_coeff(sch, t, tt, eta, ddim_inv)

        if not implicit:
            # correction evaluated at x_t (explicit form)
            if p2p:
                e_c_src = unet(xt[1:], t, txt[:1], off)
                e = unet(torch.cat([xt] * 2), t, torch.cat([unc, txt]), {"save_attn": True})
                corr = mixes(e[1:2], e_c_src, e[1:2], e[3:4])
            else:
                e = unet(torch.cat([xt[1:]] * 4), t, torch.cat([unc, txt]), off)
                corr = mixes(e[0:1], e[2:3], e[1:2], e[3:4])
            x_k = x_base + coeff * corr
        else:
            x_k = x_base.clone()
            for k in range(K):
                if p2p:
                    save = not (k < K - 1 and K > 1)
                    e_c_src = unet(x_k, tt, txt[:1], off)
                    e = unet(torch.cat([x_orig, x_k] * 2), tt, torch.cat([unc, txt]),
                             {} if masactrl else {"save_attn": save})
                    corr = mixes(e[1:2], e_c_src, e[1:2], e[3:4])
                else:
                    e = unet(torch.cat([x_k] * 4), tt, torch.cat([unc, txt]), off)
                    corr = mixes(e[0:1], e[2:3], e[1:2], e[3:4])
                rec = _l1_pull(x_k, x_base, corr, w_rec) if (k > 0 and style is None and not masactrl) else x_k
                x_k = rec + coeff * corr
                if style is not None and style[0]:
                    e_tar = e[1:2] + w_tar * (e[3:4] - e[1:2])
                    x_k = _style_step(model, style[0], x_k, e_tar, corr, tt, style[1])

        xt = torch.cat([x_orig, x_k.detach()])
        if controller is not None and not masactrl:
            xt = controller.step_callback(xt)
    return xt[1].unsqueeze(0), xt[0].unsqueeze(0)


def h_edit_r_explicit(model, xT, eta=1.0, prompts="", cfg_scales=None, zs=None, controller=None,
                      after_skip_steps=35, is_ddim_inversion=False):
    return _loop(model, xT, eta, prompts, cfg_scales, zs, controller, after_skip_steps,
                 is_ddim_inversion, p2p=False, implicit=False)


def h_edit_r_implicit(model, xT, eta=1.0, prompts="", cfg_scales=None, zs=None, controller=None,
                      weight_reconstruction=0.1, optimization_steps=1, after_skip_steps=35,
                      is_ddim_inversion=False):
    return _loop(model, xT, eta, prompts, cfg_scales, zs, controller, after_skip_steps,
                 is_ddim_inversion, p2p=False, implicit=True, K=optimization_steps,
                 w_rec=weight_reconstruction)


def h_edit_p2p_explicit(model, xT, eta=1.0, prompts="", cfg_scales=None, zs=None, controller=None,
                        is_ddim_inversion=True, after_skip_steps=35):
    return _loop(model, xT, eta, prompts, cfg_scales, zs, controller, after_skip_steps,
                 is_ddim_inversion, p2p=True, implicit=False)


def h_edit_p2p_implicit(model, xT, eta=1.0, prompts="", cfg_scales=None, zs=None, controller=None,
                        weight_reconstruction=0.075, optimization_steps=1, after_skip_steps=35,
                        is_ddim_inversion=True):
    return _loop(model, xT, eta, prompts, cfg_scales, zs, controller, after_skip_steps,
                 is_ddim_inversion, p2p=True, implicit=True, K=optimization_steps,
                 w_rec=weight_reconstruction)


def h_edit_p2p_implicit_style(model, image_encoder, xT, eta=1.0, prompts="", cfg_scales=None, zs=None,
                              controller=None, weight_edit_clip=0.55, optimization_steps=1,
                              after_skip_steps=100, is_ddim_inversion=False):
    """text-guided-n-style/inversion/h_edit.py:14-192 (its h_Edit_p2p_implicit)."""
    return _loop(model, xT, eta, prompts, cfg_scales, zs, controller, after_skip_steps,
                 is_ddim_inversion, p2p=True, implicit=True, K=optimization_steps,
                 style=(image_encoder, weight_edit_clip))


def h_edit_masactrl_implicit(model, xT, eta=0, prompts="", cfg_scales=None, zs=None, optimization_steps=1,
                             after_skip_steps=35, is_ddim_inversion=True):
    """text-guided/inversion/masactrl_h_edit.py:10-155; the editor is the one registered on model.unet."""
    return _loop(model, xT, eta, prompts, cfg_scales, zs, None, after_skip_steps, is_ddim_inversion,
                 p2p=True, implicit=True, K=optimization_steps, masactrl=True)


def h_edit_pnp_implicit(model, xT, eta=0, prompts="", cfg_scales=None, zs=None, optimization_steps=1,
                        after_skip_steps=35, is_ddim_inversion=True):
    """text-guided/inversion/pnp_h_edit.py:24-167: per step the plain 4-row base pass, then per inner step two 1-row
    passes eps(x^k, t-1, src), eps(x^k, t-1, null) and the 2-row pass [x^orig|src, x^k|tar] in which the
    registered Plug-and-Play hooks fire; no reconstruction pull, no callback."""
    from . import pnp
    sch, (w_src, w_hat, w_tar), txt, unc, xt, op = _prep(model, xT, prompts, cfg_scales, after_skip_steps)
    T = sch.num_inference_steps
    pos = {int(v): k for k, v in enumerate(op)}

    def unet(x, t, ctx):
        with torch.no_grad():
            return model.unet(x, t, encoder_hidden_states=ctx).sample

    for i, t in enumerate(op):
        idx = T - pos[int(t)] - (T - after_skip_steps + 1)
        z = zs[idx] if zs is not None else None
        tt = op[i + 1] if i < len(op) - 1 else torch.tensor(0)
        pnp.register_time(model, int(t))
        e = unet(torch.cat([xt] * 2), t, torch.cat([unc[:1], unc[:1], txt[:1], txt[:1]]))
        e_u, e_c = e.chunk(2)
        prev = S.reverse_step(sch, e_u + w_src * (e_c - e_u), t, xt, eta=eta, z=z, ddim_inv=is_ddim_inversion)
        x_orig, x_k = prev.chunk(2)
        coeff = S.edit_coeff(sch, t, tt, eta, is_ddim_inversion)
        for _ in range(optimization_steps):
            pnp.register_time(model, int(tt))
            e_c_src = unet(x_k, tt, txt[:1])
            e_u_tar = unet(x_k, tt, unc[1:])
            e_c_tar = unet(torch.cat([x_orig, x_k]), tt, txt)[1:2]
            e_hat = e_u_tar + w_hat * (e_c_src - e_u_tar)
            e_tar = e_u_tar + w_tar * (e_c_tar - e_u_tar)
            x_k = x_k + coeff * (e_tar - e_hat)
        xt = torch.cat([x_orig, x_k.detach()])
    return xt[1].unsqueeze(0), xt[0].unsqueeze(0)
