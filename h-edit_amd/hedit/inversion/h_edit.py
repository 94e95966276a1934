"""Combined text-guided + style editing -- drop-in for the reference's
text-guided-n-style/inversion/h_edit.py:14-192 (``h_Edit_p2p_implicit`` of that sub-project: note the
extra ``image_encoder`` argument in second position and ``weight_edit_clip`` in place of
``weight_reconstruction``).  Same signature, defaults, assertion and return values
(edited latent, reconstructed latent), same sequence of UNet evaluations.  The UNet passes, the
sampler arithmetic, the VAE decode and its backward run on the HIP path; ``image_encoder`` is the
caller's torch module exposing ``get_gram_matrix_residual(image)`` (the reference's
clip_guidance/base_clip.py:30-65 CLIPEncoder), differentiated by torch autograd exactly as the
reference does."""
from ..engine import HEditEngine
from .p2p_h_edit import _common


def h_Edit_p2p_implicit(model, image_encoder, xT, eta=1.0, prompts="", cfg_scales=None, prog_bar=False, zs=None,
                        controller=None, weight_edit_clip=0.55, optimization_steps=1, after_skip_steps=100,
                        is_ddim_inversion=False):
    e, x, z = _common(model, xT, eta, prompts, cfg_scales, zs)
    return HEditEngine(model).run(x, z, [prompts[:2]], cfg_scales, controller, eta=e, p2p=True, implicit=True,
                                  K=optimization_steps, after_skip_steps=after_skip_steps,
                                  ddim_inv=is_ddim_inversion, style=(image_encoder, weight_edit_clip))
