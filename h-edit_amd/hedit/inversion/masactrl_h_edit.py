"""h-Edit with MasaCtrl -- drop-in for text-guided/inversion/masactrl_h_edit.py:10-155
(``h_Edit_masactrl_implicit``): the implicit loop with the mutual-self-attention editor registered on the
model (``regiter_attention_editor_diffusers``) instead of a P2P controller; no reconstruction pull between
inner steps and no LocalBlend (masactrl_h_edit.py:139-150).  Same signature, defaults and return values."""
from ..engine import HEditEngine
from .p2p_h_edit import _common


def h_Edit_masactrl_implicit(model, xT, eta=0, prompts="", cfg_scales=None, prog_bar=False, zs=None,
                             optimization_steps=1, after_skip_steps=35, is_ddim_inversion=True):
    editor = getattr(model.unet, "_attention_editor", None)
    if editor is None:
        raise RuntimeError("register an editor first: regiter_attention_editor_diffusers(model, MutualSelfAttentionControl(...))")
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run(x, z, [prompts[:2]], cfg_scales, editor, eta=e, p2p=True, implicit=True,
                                  K=optimization_steps, after_skip_steps=after_skip_steps,
                                  ddim_inv=is_ddim_inversion, rec_pull=False)
