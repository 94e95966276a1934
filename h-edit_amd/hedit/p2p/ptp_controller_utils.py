"""Controller factory (mirrors text-guided/p2p/ptp_controller_utils.py: preprocessing :13-48,
get_equalizer :92-104, make_controller :106-134)."""
import difflib

import torch

from .ptp_classes import AttentionRefine, AttentionReplace, AttentionReweight, ControllerBatch, LocalBlend
from .ptp_utils import get_word_inds


def _differences(src_prompt, tar_prompt):
    a, b = src_prompt.split(), tar_prompt.split()
    src, tar = [], []
    for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, a, b).get_opcodes():
        if tag in ("replace", "delete"):
            src.extend(a[i1:i2])
        if tag in ("replace", "insert"):
            tar.extend(b[j1:j2])
    return " ".join(src), " ".join(tar)


def preprocessing(src_prompt, tar_prompt, is_global_edit=True, value=1.5):
    """Heuristic blend word / focus words from the prompt difference.  The reference tokenises
    with nltk.word_tokenize; plain whitespace splitting is used here (nltk is not a dependency
    of the hot path)."""
    src_text, tar_text = _differences(src_prompt, tar_prompt)
    if len(src_text) == 0 or len(tar_text) == 0 or not is_global_edit:
        blend_word = None
    else:
        blend_word = ((src_text,), (tar_text,))
    focus = tar_text.split()
    eq_params = {"words": tuple(focus), "values": tuple(value for _ in focus)} if focus else None
    return blend_word, eq_params


def preprocessing_attn_focus(src_prompt, tar_prompt, is_global_edit=True):
    return preprocessing(src_prompt, tar_prompt, is_global_edit, value=1.25)


def get_equalizer(text, word_select, values, tokenizer):
    if isinstance(word_select, (int, str)):
        word_select = (word_select,)
    equalizer = torch.ones(1, 77)
    for word, val in zip(word_select, values):
        equalizer[:, get_word_inds(text, word, tokenizer)] = val
    return equalizer


def make_controller(prompts, is_replace_controller, cross_replace_steps, self_replace_steps,
                    blend_word=None, equilizer_params=None, num_steps=None, tokenizer=None, device=None):
    lb = None if blend_word is None else LocalBlend(prompts, num_steps, blend_word, tokenizer=tokenizer, device=device)
    cls = AttentionReplace if is_replace_controller else AttentionRefine
    controller = cls(prompts, num_steps, cross_replace_steps=cross_replace_steps,
                     self_replace_steps=self_replace_steps, local_blend=lb, tokenizer=tokenizer, device=device)
    if equilizer_params is not None:
        eq = get_equalizer(prompts[1], equilizer_params["words"], equilizer_params["values"], tokenizer=tokenizer)
        controller = AttentionReweight(prompts, num_steps, cross_replace_steps=cross_replace_steps,
                                       self_replace_steps=self_replace_steps, equalizer=eq, local_blend=lb,
                                       controller=controller, tokenizer=tokenizer, device=device)
    return controller


def make_controller_batch(specs, **common):
    """specs: list of dicts with the per-image arguments of make_controller (prompts,
    is_replace_controller, blend_word, equilizer_params); `common` holds the shared ones."""
    return ControllerBatch([make_controller(**{**common, **s}) for s in specs])
