"""h-Edit with Plug-and-Play -- drop-in for text-guided/inversion/pnp_h_edit.py:24-167 (``h_Edit_PnP_implicit``; its
``register_time`` twin lives in hedit.plug_n_play.pnp_utils).  Same signature, defaults and return values; the
injection schedules are the ones registered on the model with register_attention_control_efficient /
register_conv_control_efficient."""
from ..engine import HEditEngine
from ..plug_n_play.pnp_utils import register_time  # noqa: F401  (the reference module defines it too, :7-22)
from .p2p_h_edit import _common


def h_Edit_PnP_implicit(model, xT, eta=0, prompts="", cfg_scales=None, prog_bar=False, zs=None,
                        optimization_steps=1, after_skip_steps=35, is_ddim_inversion=True):
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run_pnp(x, None if z is None else z[:, 0], prompts[:2], cfg_scales, eta=e, K=optimization_steps,
                                      after_skip_steps=after_skip_steps, ddim_inv=is_ddim_inversion)
