"""h-Edit sampling loops -- drop-in for text-guided/inversion/p2p_h_edit.py:
h_Edit_R_explicit :21-156, h_Edit_R_implicit :162-362, h_Edit_p2p_explicit :380-523,
h_Edit_p2p_implicit :529-701.  Same signatures, defaults, assertions and return values
(edited latent, reconstructed latent), same sequence of UNet evaluations; the work runs on the
HIP path (hedit.engine.HEditEngine with one image)."""
from ..engine import HEditEngine


def _common(model, xT, eta, prompts, cfg_scales, zs):
    batch_size = len(prompts)
    assert batch_size >= 2, "only support prompt editing"
    if type(eta) in [int, float]:
        etas = [eta] * model.scheduler.num_inference_steps
    else:
        etas = eta
    assert len(etas) == model.scheduler.num_inference_steps
    assert len(cfg_scales) == 3, "cfg_scales = [w_src, w_src_edit, w_tar]"
    x = xT.unsqueeze(0) if xT.dim() < 4 else xT
    z = None if zs is None else zs[:, None]          # (T',1,C,H,W)
    etas = [float(e) for e in etas]
    return (etas[0] if all(e == etas[0] for e in etas) else etas), x, z


def h_Edit_R_explicit(model, xT, eta=1.0, prompts="", cfg_scales=None, prog_bar=False, zs=None, controller=None,
                      after_skip_steps=35, is_ddim_inversion=False):
    assert (len(prompts) >= 2) and (not is_ddim_inversion), "only support prompt editing and DDPM sampling"
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run(x, z, [prompts[:2]], cfg_scales, controller, eta=e, p2p=False, implicit=False,
                                  after_skip_steps=after_skip_steps, ddim_inv=is_ddim_inversion)


def h_Edit_R_implicit(model, xT, eta=1.0, prompts="", cfg_scales=None, prog_bar=False, zs=None, controller=None,
                      weight_reconstruction=0.1, optimization_steps=1, after_skip_steps=35,
                      is_ddim_inversion=False):
    assert (len(prompts) >= 2) and (not is_ddim_inversion), "only support prompt editing and DDPM sampling"
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run(x, z, [prompts[:2]], cfg_scales, controller, eta=e, p2p=False, implicit=True,
                                  K=optimization_steps, w_rec=weight_reconstruction,
                                  after_skip_steps=after_skip_steps, ddim_inv=is_ddim_inversion)


def h_Edit_p2p_explicit(model, xT, eta=1.0, prompts="", cfg_scales=None, prog_bar=False, zs=None, controller=None,
                        is_ddim_inversion=True, after_skip_steps=35):
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run(x, z, [prompts[:2]], cfg_scales, controller, eta=e, p2p=True, implicit=False,
                                  after_skip_steps=after_skip_steps, ddim_inv=is_ddim_inversion)


def h_Edit_p2p_implicit(model, xT, eta=1.0, prompts="", cfg_scales=None, prog_bar=False, zs=None, controller=None,
                        weight_reconstruction=0.075, optimization_steps=1, after_skip_steps=35,
                        is_ddim_inversion=True):
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run(x, z, [prompts[:2]], cfg_scales, controller, eta=e, p2p=True, implicit=True,
                                  K=optimization_steps, w_rec=weight_reconstruction,
                                  after_skip_steps=after_skip_steps, ddim_inv=is_ddim_inversion)
