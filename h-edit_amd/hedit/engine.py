"""Batched h-Edit sampler on the HIP path.

One engine runs n independent images in lock-step (the reference edits one image at a time,
text-guided/main_p2p.py:110; images are independent, so batching is a pure throughput
generalisation -- SURVEY.md section 0.1 / 8e).  For n = 1 the sequence of UNet evaluations is the
reference's, call for call (p2p_h_edit.py:613,644,652 etc.):

    per step:  1 base pass (4n rows, controller off)
               K x [ 1 source pass (n rows, controller off) + 1 P2P pass (4n rows, controller on) ]

Row layout of every batched tensor is [row-kind][image]: e.g. the P2P pass feeds
[x_orig|null]*n, [x_k|null]*n, [x_orig|src]*n, [x_k|tar]*n, which for n = 1 is exactly the
reference's batch.  All arithmetic between UNet calls runs in the fused HIP step kernels
(csrc/step.hip); the host only computes the scalar schedule coefficients (fp32, same formulas
as inversion_utils.py:38-56,84-119,168-195) and never synchronises with the device.
"""
import ctypes as C

import torch

from . import _lib


def _f(x):
    return float(x)


class Schedule:
    """Scalar coefficient tables of a DDIM scheduler (host, fp32 like the reference)."""

    def __init__(self, scheduler):
        self.s = scheduler
        self.ab = scheduler.alphas_cumprod.float().cpu()
        self.final = scheduler.final_alpha_cumprod.float().cpu() if isinstance(
            scheduler.final_alpha_cumprod, torch.Tensor) else torch.tensor(float(scheduler.final_alpha_cumprod))
        self.T = scheduler.num_inference_steps
        self.n_train = scheduler.config.num_train_timesteps

    def ab_prev(self, t):
        p = t - self.n_train // self.T
        return self.ab[p] if p >= 0 else self.final

    def variance(self, t):
        a_t, a_p = self.ab[t], self.ab_prev(t)
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def full_coeff(self, t, t_prev, eta, ddim_inv):
        ab = self.ab
        sig, a = (1 - ab) ** 0.5, ab ** 0.5
        omega = eta * (sig[t_prev] / (sig[t] * a[t_prev])) * ((ab[t_prev] - ab[t]) ** 0.5)
        if ddim_inv:
            omega = 0
        return (1 - ab[t_prev] - omega ** 2) ** 0.5

    def edit_coeff(self, t, t_prev, eta, ddim_inv):
        ab = self.ab
        ratio = (ab ** 0.5)[t_prev] / (ab ** 0.5)[t]
        return self.full_coeff(t, t_prev, eta, ddim_inv) - ((1 - ab) ** 0.5)[t] * ratio

    def step_coef(self, t, t_prev, eta, ddim_inv, cfg, w_rec=0.0, coeff=None):
        a_t, a_p = self.ab[t], self.ab_prev(t)
        var = self.variance(t)
        c = _lib.StepCoef()
        c.sqrt_ab_t = _f(a_t ** 0.5)
        c.sqrt_1m_ab_t = _f((1 - a_t) ** 0.5)
        c.sqrt_ab_prev = _f(a_p ** 0.5)
        c.dir_coef = _f((1 - a_p) ** 0.5) if ddim_inv else _f((1 - a_p - (eta ** 2) * var) ** 0.5)
        c.noise_coef = 0.0 if eta <= 0 else (_f(eta) if ddim_inv else _f(eta * var ** 0.5))
        c.w_src, c.w_hat, c.w_tar = (float(v) for v in cfg)
        c.coeff = _f(self.edit_coeff(t, t_prev, eta, ddim_inv)) if coeff is None else _f(coeff)
        c.w_rec = float(w_rec)
        return c


class HEditEngine:
    def __init__(self, model):
        self.model = model
        self.unet = model.unet
        self.lib = _lib.lib()
        self.dev = model.unet.device
        self.style_chunk = 8      # images per decoder forward + backward inside style_step (~1.1 GiB of tape each)

    # ------------------------------------------------------------------ kernels
    def step_base(self, eps, xt, z, out, n, rows, coef):
        elems = xt[0].numel()
        _lib.check(self.lib.hedit_step_base(_lib.ptr(eps), _lib.ptr(xt), _lib.ptr(z), _lib.ptr(out), n, elems,
                                            rows, C.byref(coef), _lib.cur_stream()))

    def step_update(self, e_u_src, e_c_src, e_u_tar, e_c_tar, x_k, x_base, out, n, k_gt0, coef):
        elems = x_k[0].numel()
        for t_ in (e_u_src, e_c_src, e_u_tar, e_c_tar, x_k, x_base, out):
            assert t_.is_contiguous() and t_.dtype == torch.float32
        _lib.check(self.lib.hedit_step_update(_lib.ptr(e_u_src), _lib.ptr(e_c_src), _lib.ptr(e_u_tar),
                                              _lib.ptr(e_c_tar), elems, _lib.ptr(x_k), _lib.ptr(x_base),
                                              _lib.ptr(out), n, elems, int(k_gt0), C.byref(coef),
                                              _lib.cur_stream()))

    def style_step(self, e_u_src, e_c_src, e_u_tar, e_c_tar, x, tt, cfg_scales, image_encoder, weight):
        """Style-guidance update of x_{t-1}^k, text-guided-n-style/inversion/h_edit.py:160-188: Tweedie
        x0 at t-1 (HIP) -> vae.decode (HIP, an autograd node whose backward is hedit_vae_decode_vjp) ->
        the caller's image encoder (a torch module, like the text encoder it stays on PyTorch-ROCm) ->
        ||Gram residual|| -> d/dz0 -> x - rho g (HIP).  ``image_encoder``: one object, or one per image
        (its get_gram_matrix_residual reads batch item 0 only, clip_guidance/base_clip.py:60-65)."""
        vae = self.model.vae
        if vae is None:
            raise RuntimeError("style guidance needs model.vae (hedit.vae.AutoencoderKL)")
        n, elems = x.shape[0], x[0].numel()
        ab = float(self.model.scheduler.alphas_cumprod[tt])
        sab, s1m = ab ** 0.5, (1.0 - ab) ** 0.5
        inv_scale = 1.0 / 0.18215
        for t_ in (e_u_src, e_c_src, e_u_tar, e_c_tar, x):
            assert t_.is_contiguous() and t_.dtype == torch.float32
        z0 = torch.empty_like(x)
        _lib.check(self.lib.hedit_step_tweedie(_lib.ptr(e_u_tar), _lib.ptr(e_c_tar), elems, _lib.ptr(x), _lib.ptr(z0),
                                               n, elems, float(cfg_scales[2]), sab, s1m, inv_scale, _lib.cur_stream()))
        shared = not isinstance(image_encoder, (list, tuple))
        encs = [image_encoder] * n if shared else image_encoder
        g_z = torch.empty_like(x)
        with torch.enable_grad():
            # the images are independent: decode a chunk in one pass, one loss per image, and the gradient of
            # the SUM w.r.t. the chunk's latents is the stack of the per-image gradients
            for lo in range(0, n, self.style_chunk):
                hi = min(n, lo + self.style_chunk)
                zc = z0[lo:hi].clone().requires_grad_(True)
                img = vae.decode(zc).sample
                if shared and hasattr(image_encoder, "gram_residual_norms"):
                    loss = image_encoder.gram_residual_norms(img).sum()          # native on the GPU (csrc/vit.hip)
                elif (not shared) and all(hasattr(e, "gram_residual_norms_each") for e in encs[lo:hi]):
                    loss = encs[lo].gram_residual_norms_each(encs[lo:hi], img).sum()     # one call, one reference per image
                elif shared and hasattr(image_encoder, "gram_residuals"):
                    loss = torch.linalg.norm(image_encoder.gram_residuals(img), dim=(1, 2)).sum()
                else:
                    loss = sum(torch.linalg.norm(encs[i].get_gram_matrix_residual(img[i - lo:i - lo + 1]))
                               for i in range(lo, hi))
                g_z[lo:hi] = torch.autograd.grad(outputs=loss, inputs=zc)[0]
        out = torch.empty_like(x)
        _lib.check(self.lib.hedit_step_style(_lib.ptr(e_u_src), _lib.ptr(e_c_src), _lib.ptr(e_u_tar), _lib.ptr(e_c_tar),
                                             elems, _lib.ptr(x), _lib.ptr(g_z), _lib.ptr(out), n, elems,
                                             float(cfg_scales[1]), float(cfg_scales[2]), inv_scale / sab, float(weight),
                                             _lib.cur_stream()))
        return out

    # ------------------------------------------------------------------ text
    def encode(self, prompts):
        """Prompt by prompt, each as a batch of one: the text encoder is a torch module whose GEMMs pick batch-dependent
        kernels, and an image's embedding must not depend on the prompts it happens to be batched with (the engine's
        results are bit-identical across batch sizes, tests/test_gpu_invariance.py)."""
        out = []
        for p in prompts:
            tok = self.model.tokenizer([p], padding="max_length", max_length=self.model.tokenizer.model_max_length,
                                       truncation=True, return_tensors="pt")
            with torch.no_grad():
                out.append(self.model.text_encoder(tok.input_ids.to(self.dev))[0].float())
        return torch.cat(out)

    # ------------------------------------------------------------------ the loop
    @torch.no_grad()
    def run(self, xT, zs, prompt_pairs, cfg_scales, controller=None, eta=1.0, p2p=True, implicit=True,
            K=1, w_rec=0.1, after_skip_steps=None, ddim_inv=False, ctx=None, fuse_src_pass=False,
            reuse_orig_eps=False, style=None, rec_pull=True):
        """xT: (n,C,H,W); zs: (T',n,C,H,W) or None; prompt_pairs: n x [src, tar].
        ctx: optional precomputed (null, src, tar) embeddings ((1|n,77,D), (n,77,D), (n,77,D)).
        fuse_src_pass: evaluate eps(x^k, t-1, src) (the reference's separate n-row call,
        p2p_h_edit.py:644) as n extra un-edited rows of the P2P pass (5n rows): same arithmetic and
        FLOPs, one UNet launch sequence fewer per inner step.
        reuse_orig_eps (implicit P2P loop only, OFF by default): the P2P pass at t-1 evaluates
        eps(x^orig_{t-1}, t-1, null) and eps(x^orig_{t-1}, t-1, src) -- rows the controller never
        edits (ptp_classes.py:96-98,213-220) -- and the reference's next base pass
        (p2p_h_edit.py:604-616) recomputes exactly these two.  Reusing them evaluates 2n instead of
        4n rows in every base pass but the first: 7 instead of 9 sample-forwards per step at K=1.
        Same mathematics, fewer UNet evaluations than the reference issues.
        style = (image_encoder | None, weight_edit_clip) selects the text + style loop of
        text-guided-n-style/inversion/h_edit.py (implicit P2P only): no reconstruction pull between
        inner steps (h_edit.py:149) and one style_step after every text update (h_edit.py:160-188).
        rec_pull=False drops the L1 reconstruction pull of inner steps k > 0 (the MasaCtrl loop,
        masactrl_h_edit.py:139-150); ``controller`` may be any object with _plan / _after_pass /
        step_callback (P2P controllers, hedit.masactrl.MutualSelfAttentionControl).
        eta: one number, or the reference's per-step list etas[idx] (len == num_inference_steps, indexed like zs).
        Returns (edit (n,C,H,W), recon (n,C,H,W))."""
        if style is not None and not (p2p and implicit):
            raise ValueError("style guidance is defined for the implicit P2P loop only (n-style h_edit.py)")
        sch = self.model.scheduler
        S = Schedule(sch)
        T = sch.num_inference_steps
        if after_skip_steps is None:
            after_skip_steps = T
        n = xT.shape[0]
        dev = self.dev
        xT = xT.to(device=dev, dtype=torch.float32)
        if zs is not None:
            zs = zs.to(device=dev, dtype=torch.float32).contiguous()
        if ctx is None:
            null = self.encode([""])
            src = self.encode([p[0] for p in prompt_pairs])
            tar = self.encode([p[1] for p in prompt_pairs])
        else:
            null, src, tar = (c.to(device=dev, dtype=torch.float32) for c in ctx)
        nulln = null.expand(n, -1, -1) if null.shape[0] == 1 else null
        ctx_base4 = torch.cat([nulln, nulln, src, src]).contiguous()
        ctx_base2 = torch.cat([nulln, src]).contiguous()
        ctx_edit = torch.cat([nulln, nulln, src, tar]).contiguous()
        ctx_src = src.contiguous()
        ctx_edit5 = torch.cat([nulln, nulln, src, tar, src]).contiguous() if fuse_src_pass else None

        ts = [int(v) for v in sch.timesteps]
        op = ts[-after_skip_steps:]
        ahead = ts[-(after_skip_steps + 1)] if after_skip_steps != T else -1
        xt = torch.cat([xT, xT]).contiguous()          # [x_orig]*n, [x_edit]*n
        x_prev = torch.empty_like(xt)
        off = None

        from .p2p.ptp_classes import runs_in_python
        foreign = runs_in_python(controller)
        if foreign:
            # a host-language controller (reference protocol, ptp_classes.py:91-108) sees the reference's batch: one image,
            # rows [null, null, src, tar] -- it slices attn[h // 2:] and reshapes by its own batch_size
            if n != 1 or fuse_src_pass or reuse_orig_eps:
                raise ValueError("a foreign controller runs one image at a time, without fuse_src_pass / reuse_orig_eps "
                                 "(it sees the reference's 4-row batch)")
            if not callable(controller):
                raise TypeError("a foreign controller must be callable as controller(attn, is_cross, place_in_unet, save_attn)")
        # (decided before the first -- slow, hooked -- sampler step: a plain callable without the reference's step_callback
        #  is a controller that does nothing between steps, not an AttributeError after the first step)
        step_cb = getattr(controller, "step_callback", None) if controller is not None else None
        if step_cb is not None and not callable(step_cb):
            raise TypeError("controller.step_callback must be callable as step_callback(x_t) -> x_t")

        def p2p_pass(x_in, t, save):
            rows = x_in.shape[0]
            if foreign:
                return self.unet.forward_hooked(x_in, t, ctx_edit, controller, save)
            # controller=None: the reference's processors then leave every attention map alone (ptp_utils.py:98-101)
            plan = controller._plan(self.unet, rows, x_in.shape[2], x_in.shape[3], save) if controller is not None else None
            e = self.unet.forward_raw(x_in, t, ctx_edit5 if rows == 5 * n else ctx_edit, plan)
            if controller is not None:
                controller._after_pass(save)
            return e

        carry = None      # (eps(x_orig, t, null), eps(x_orig, t, src)) from the previous P2P pass
        reuse = reuse_orig_eps and p2p and implicit
        for i, t in enumerate(op):
            idx = T - i - (T - after_skip_steps + 1)
            z = zs[idx] if zs is not None else None
            tt = op[i + 1] if i < len(op) - 1 else 0
            eta_i = float(eta[idx]) if isinstance(eta, (list, tuple)) else float(eta)   # etas[idx], p2p_h_edit.py:619,665
            coef = S.step_coef(t, tt, eta_i, ddim_inv, cfg_scales, w_rec)

            if (not p2p) and implicit and i == 0 and ahead != -1:
                # one extra correction of the start sample when steps were skipped (p2p_h_edit.py:239-267)
                xe = xt[n:]
                e = self.unet.forward_raw(torch.cat([xe] * 4), t, ctx_edit, off)
                c0 = S.step_coef(t, tt, eta_i, ddim_inv, cfg_scales, w_rec, coeff=S.edit_coeff(ahead, t, eta_i, ddim_inv))
                new = torch.empty_like(xe)
                self.step_update(e[0:n], e[2 * n:3 * n], e[n:2 * n], e[3 * n:], xe, xe, new, n, False, c0)
                xt = torch.cat([xt[:n], new]).contiguous()

            # ---- base pass -> x_{t-1}^orig, x_{t-1}^base
            if p2p and reuse and carry is not None:
                e2 = self.unet.forward_raw(torch.cat([xt[n:], xt[n:]]), t, ctx_base2, off)
                e = torch.cat([carry[0], e2[:n], carry[1], e2[n:]])
                self.step_base(e, xt, z, x_prev, n, 4, coef)
            elif p2p:
                e = self.unet.forward_raw(torch.cat([xt, xt]), t, ctx_base4, off)
                self.step_base(e, xt, z, x_prev, n, 4, coef)
            else:
                e = self.unet.forward_raw(torch.cat([xt[n:], xt[n:]]), t, ctx_base2, off)
                self.step_base(e, xt, z, x_prev, n, 2, coef)
            x_orig, x_base = x_prev[:n], x_prev[n:]

            if not implicit:
                new = torch.empty_like(x_base)
                if p2p:
                    e_src = self.unet.forward_raw(xt[n:].contiguous(), t, ctx_src, off)
                    e = p2p_pass(torch.cat([xt, xt]), t, True)
                    self.step_update(e[n:2 * n], e_src, e[n:2 * n], e[3 * n:], x_base, x_base, new, n, False, coef)
                else:
                    e = self.unet.forward_raw(torch.cat([xt[n:]] * 4), t, ctx_edit, off)
                    self.step_update(e[0:n], e[2 * n:3 * n], e[n:2 * n], e[3 * n:], x_base, x_base, new, n, False, coef)
                x_k = new
            else:
                x_k = x_base.clone()
                for k in range(K):
                    new = torch.empty_like(x_k)
                    if p2p:
                        save = not (k < K - 1 and K > 1)
                        if fuse_src_pass:
                            e = p2p_pass(torch.cat([x_orig, x_k, x_orig, x_k, x_k]), tt, save)
                            e_src = e[4 * n:]
                        else:
                            e_src = self.unet.forward_raw(x_k, tt, ctx_src, off)
                            e = p2p_pass(torch.cat([x_orig, x_k, x_orig, x_k]), tt, save)
                        self.step_update(e[n:2 * n], e_src, e[n:2 * n], e[3 * n:4 * n], x_k, x_base, new, n,
                                         k > 0 and style is None and rec_pull, coef)
                        if style is not None and style[0] is not None:
                            with torch.enable_grad():
                                new = self.style_step(e[n:2 * n], e_src, e[n:2 * n], e[3 * n:4 * n], new, tt, cfg_scales,
                                                      style[0], style[1])
                        if reuse:
                            carry = (e[0:n], e[2 * n:3 * n])
                    else:
                        e = self.unet.forward_raw(torch.cat([x_k] * 4), tt, ctx_edit, off)
                        self.step_update(e[0:n], e[2 * n:3 * n], e[n:2 * n], e[3 * n:], x_k, x_base, new, n, k > 0, coef)
                    x_k = new

            xt = torch.cat([x_orig, x_k]).contiguous()
            if step_cb is not None:
                xt = step_cb(xt)
        return xt[n:].clone(), xt[:n].clone()

    # ------------------------------------------------------------------ Plug-and-Play loop
    @torch.no_grad()
    def run_pnp(self, xT, zs, prompts, cfg_scales, eta=1.0, K=1, after_skip_steps=None, ddim_inv=True):
        """h-Edit with Plug-and-Play injection, text-guided/inversion/pnp_h_edit.py:24-167, for n images in lock-step
        (the reference edits one: its injection rule only fires for a batch of two).  xT (n,C,H,W); zs (T',n,C,H,W) or
        (T',C,H,W) for one image; prompts [src, tar] or n x [src, tar].  Per step: the 4n-row base pass (plain; with four
        rows the hooks stay silent, pnp_utils.py:46-50), then per inner step eps(x^k, t-1, null) and eps(x^k, t-1, src)
        -- two 1-row calls in the reference, one plain 2n-row call here -- and the 2n-row pass
        [x^orig | src] * n, [x^k | tar] * n under the registered injection plan (row n + i takes row i's q, k /
        features).  Returns (edit, recon)."""
        from .plug_n_play.pnp_utils import register_time
        sch = self.model.scheduler
        S = Schedule(sch)
        T = sch.num_inference_steps
        if after_skip_steps is None:
            after_skip_steps = T
        dev = self.dev
        xT = xT.to(device=dev, dtype=torch.float32)
        n = xT.shape[0]
        pairs = [list(prompts[:2])] if isinstance(prompts[0], str) else [list(p[:2]) for p in prompts]
        if len(pairs) != n:
            raise ValueError("one [source, target] prompt pair per image expected")
        if zs is not None:
            zs = zs.to(device=dev, dtype=torch.float32).contiguous()
            if zs.dim() == 4:
                zs = zs[:, None]
        null = self.encode([""]).expand(n, -1, -1)
        src, tar = self.encode([p[0] for p in pairs]), self.encode([p[1] for p in pairs])
        ctx_base4 = torch.cat([null, null, src, src]).contiguous()
        ctx_k2 = torch.cat([null, src]).contiguous()
        ctx_pair = torch.cat([src, tar]).contiguous()
        editor = getattr(self.unet, "_attention_editor", None)
        ts = [int(v) for v in sch.timesteps]
        op = ts[-after_skip_steps:]
        xt = torch.cat([xT, xT]).contiguous()
        x_prev = torch.empty_like(xt)
        H, W = xT.shape[2], xT.shape[3]
        for i, t in enumerate(op):
            idx = T - i - (T - after_skip_steps + 1)
            z = zs[idx] if zs is not None else None
            tt = op[i + 1] if i < len(op) - 1 else 0
            eta_i = float(eta[idx]) if isinstance(eta, (list, tuple)) else float(eta)
            coef = S.step_coef(t, tt, eta_i, ddim_inv, cfg_scales, 0.0)
            register_time(self.model, t)
            e = self.unet.forward_raw(torch.cat([xt, xt]), t, ctx_base4, None)
            self.step_base(e, xt, z, x_prev, n, 4, coef)
            x_orig, x_base = x_prev[:n], x_prev[n:]
            x_k = x_base.clone()
            for _ in range(K):
                register_time(self.model, tt)
                e2 = self.unet.forward_raw(torch.cat([x_k, x_k]), tt, ctx_k2, None)
                plan = editor._plan(self.unet, 2 * n, H, W, True, n_images=n) if editor is not None else None
                ep = self.unet.forward_raw(torch.cat([x_orig, x_k]), tt, ctx_pair, plan)
                if editor is not None:
                    editor._after_pass(True)
                new = torch.empty_like(x_k)
                self.step_update(e2[0:n], e2[n:2 * n], e2[0:n], ep[n:2 * n].contiguous(), x_k, x_base, new, n, False, coef)
                x_k = new
            xt = torch.cat([x_orig, x_k]).contiguous()
        return xt[n:].clone(), xt[:n].clone()

    # ------------------------------------------------------------------ DDPM inversion
    @torch.no_grad()
    def ddpm_inversion(self, x0, prompts, eta=1.0, cfg_src=1.0, noise=None, generator=None, return_noise=False):
        """Edit-friendly DDPM inversion for n images (ddpm_inversion.py:54-167).
        x0 (n,C,H,W); prompts: n source prompts ("" = unconditional: its embedding IS the null embedding, so the
        CFG mix of such a row reduces to eps(null) exactly, per image, as the reference decides per image).
        noise: optional (T+1,n,C,H,W) tensor of the forward-process noises (row idx as in the reference).
        Each step runs in hedit_step_invert with the arithmetic of the sampler's base step, so the loop's
        reconstruction branch retraces xts bit for bit (ddpm_inversion.py:146-162 <-> inversion_utils.py:84-119).
        Returns zs (T,n,C,H,W), xts (T+1,n,C,H,W) [, noise_added (T+1,n,C,H,W) if return_noise]."""
        sch = self.model.scheduler
        S = Schedule(sch)
        T = sch.num_inference_steps
        dev = self.dev
        x0 = x0.to(device=dev, dtype=torch.float32)
        n = x0.shape[0]
        elems = x0[0].numel()
        ts = [int(v) for v in sch.timesteps]
        ab = S.ab
        xts = torch.zeros(T + 1, *x0.shape, device=dev)
        nzs = torch.zeros(T + 1, *x0.shape, device=dev) if return_noise else None
        xts[0] = x0
        for j, t in enumerate(reversed(ts)):
            idx = j + 1
            if noise is not None:
                nz = noise[idx].to(dev)
            else:
                nz = torch.randn(x0.shape, device=dev, generator=generator)
            if nzs is not None:
                nzs[idx] = nz
            xts[idx] = x0 * _f(ab[t] ** 0.5) + nz * _f((1 - ab[t]) ** 0.5)
        null = self.encode([""]).expand(n, -1, -1)
        cond = any(p != "" for p in prompts)
        ctx = torch.cat([null, self.encode(list(prompts))]).contiguous() if cond else null.contiguous()
        zs = torch.zeros(T, *x0.shape, device=dev)
        for i, t in enumerate(ts):
            idx = T - i - 1
            xt = xts[idx + 1]
            if cond:
                e = self.unet.forward_raw(torch.cat([xt, xt]), t, ctx)
                e_u, e_c = e[:n], e[n:]
            else:
                e_u = e_c = self.unet.forward_raw(xt.contiguous(), t, ctx)
            eta_i = float(eta[idx]) if isinstance(eta, (list, tuple)) else float(eta)      # etas[idx], ddpm_inversion.py:152-161
            coef = S.step_coef(t, 0, eta_i, False, (cfg_src, 0.0, 0.0), coeff=0.0)
            _lib.check(self.lib.hedit_step_invert(_lib.ptr(e_u), _lib.ptr(e_c), _lib.ptr(xt), _lib.ptr(xts[idx]),
                                                  _lib.ptr(zs[idx]), n, elems, C.byref(coef), _lib.cur_stream()))
        if return_noise:
            return zs, xts, nzs
        return zs, xts

    # ------------------------------------------------------------------ DDIM inversion (h-Edit-D)
    @torch.no_grad()
    def ddim_inversion(self, w0, prompts, cfg_scale):
        """Deterministic DDIM inversion for n images + the per-step corrections u_t replayed by
        h-Edit-D (text-guided/inversion/ddim_inversion.py:54-131; two passes of T UNet calls with
        2n rows each).  w0 (n,C,H,W).  Returns (latent_T (n,..), zs (T,n,..), latents (T+1,n,..))."""
        sch = self.model.scheduler
        S = Schedule(sch)
        T = sch.num_inference_steps
        step = S.n_train // T
        dev = self.dev
        lat = w0.to(device=dev, dtype=torch.float32).clone()
        n = lat.shape[0]
        ts = [int(v) for v in sch.timesteps]
        null = self.encode([""]).expand(n, -1, -1)
        ctx = torch.cat([null, self.encode(list(prompts))]).contiguous()
        ab = S.ab

        def eps(x, t):
            e = self.unet.forward_raw(torch.cat([x, x]), t, ctx)
            return e[:n] + cfg_scale * (e[n:] - e[:n])

        lats = torch.zeros(T + 1, *lat.shape, device=dev)
        lats[0] = lat
        for i in range(T):
            t = ts[len(ts) - i - 1]
            e = eps(lat, t)
            cur = min(t - step, 999)
            a_cur = ab[cur] if cur >= 0 else S.final
            a_next = ab[t]
            x0 = (lat - _f((1 - a_cur) ** 0.5) * e) / _f(a_cur ** 0.5)
            lat = _f(a_next ** 0.5) * x0 + _f((1 - a_next) ** 0.5) * e
            lats[i + 1] = lat
        zs = torch.zeros(T, *lat.shape, device=dev)
        for i, t in enumerate(ts):
            idx = T - i - 1
            xt = lats[idx + 1]
            e = eps(xt, t)
            x0 = (xt - _f((1 - ab[t]) ** 0.5) * e) / _f(ab[t] ** 0.5)
            a_p = S.ab_prev(t)
            mu = _f(a_p ** 0.5) * x0 + _f((1 - a_p) ** 0.5) * e
            z = lats[idx] - mu
            zs[idx] = z
            lats[idx] = mu + z
        return lat, zs, lats
