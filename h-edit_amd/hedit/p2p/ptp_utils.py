"""Registration + schedule tables of the P2P plug-in (host side).

Mirrors the parts of text-guided/p2p/ptp_utils.py that are on the h-Edit path:
P2PCrossAttnProcessor (:31-122), register_attention_control (:277-295), get_word_inds (:297-315),
update_alpha_time_word (:318-328), get_time_words_attention_alpha (:331-349).  The attention body
of the processor is NOT here: it runs inside the HIP kernels (csrc/attn.hip); the processor object
only tells the UNet which controller drives the edit and where the layer sits.  A controller that is not
one of hedit's (any callable with the reference's signature) is called through the executor's hook on
materialised probabilities (hedit/unet.py::forward_hooked).
"""
import torch

from .seq_aligner import get_word_inds  # noqa: F401  (re-exported like the reference)


class P2PCrossAttnProcessor:
    def __init__(self, controller, place_in_unet):
        self.controller = controller
        self.place_in_unet = place_in_unet

    def __call__(self, *a, **k):
        raise RuntimeError("the attention body of P2PCrossAttnProcessor is the HIP kernels (fused), or the executor's hook "
                           "path for a host-language controller (hedit.unet.UNet2DConditionModel.forward_hooked); the "
                           "processor object itself is a marker and cannot be called")


def register_attention_control(model, controller):
    procs = {}
    for name in model.unet.attn_processors.keys():
        if name.startswith("mid_block"):
            place = "mid"
        elif name.startswith("up_blocks"):
            place = "up"
        elif name.startswith("down_blocks"):
            place = "down"
        else:
            continue
        procs[name] = P2PCrossAttnProcessor(controller=controller, place_in_unet=place)
    model.unet.set_attn_processor(procs)
    controller.num_att_layers = len(procs)


def update_alpha_time_word(alpha, bounds, prompt_ind, word_inds=None):
    if isinstance(bounds, float):
        bounds = 0, bounds
    start, end = int(bounds[0] * alpha.shape[0]), int(bounds[1] * alpha.shape[0])
    if word_inds is None:
        word_inds = torch.arange(alpha.shape[2])
    alpha[:start, prompt_ind, word_inds] = 0
    alpha[start:end, prompt_ind, word_inds] = 1
    alpha[end:, prompt_ind, word_inds] = 0
    return alpha


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    if not isinstance(cross_replace_steps, dict):
        cross_replace_steps = {"default_": cross_replace_steps}
    if "default_" not in cross_replace_steps:
        cross_replace_steps["default_"] = (0.0, 1.0)
    n_edit = len(prompts) - 1
    table = torch.zeros(num_steps + 1, n_edit, max_num_words)
    for i in range(n_edit):
        table = update_alpha_time_word(table, cross_replace_steps["default_"], i)
    for word, bounds in cross_replace_steps.items():
        if word == "default_":
            continue
        for i in range(n_edit):
            ind = get_word_inds(prompts[i + 1], word, tokenizer)
            if len(ind) > 0:
                table = update_alpha_time_word(table, bounds, i, torch.as_tensor(ind))
    return table.reshape(num_steps + 1, n_edit, 1, 1, max_num_words)


@torch.no_grad()
def latent2image(vae, latents):
    """latents -> uint8 HWC numpy images (reference ptp_utils.py:181-187): decode(latents / 0.18215),
    (x / 2 + 0.5).clamp(0, 1) * 255."""
    image = vae.decode(1 / 0.18215 * latents)["sample"]
    image = (image / 2 + 0.5).clamp(0, 1)
    return (image.cpu().permute(0, 2, 3, 1).numpy() * 255).astype("uint8")
