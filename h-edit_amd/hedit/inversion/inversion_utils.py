"""Scheduler algebra + text encoding with the reference's function names
(text-guided/inversion/inversion_utils.py: encode_text :13-35, get_variance :38-56,
reverse_step :58-126, reverse_step_pred_x0 :128-140, compute_full_coeff :168-195).

Scalars are computed on the host in fp32 exactly like the reference; the tensor part of
``reverse_step`` runs in the fused HIP step kernel when called through the h_Edit loops.  The
functions below operate on whatever device their tensor arguments live on and exist so that
driver code written against the reference keeps working.
"""
import torch

from ..engine import Schedule


def encode_text(model, prompts):
    text_input = model.tokenizer(prompts, padding="max_length", max_length=model.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt")
    with torch.no_grad():
        return model.text_encoder(text_input.input_ids.to(model.device))[0]


def get_variance(model, timestep):
    return Schedule(model.scheduler).variance(int(timestep))


def reverse_step_pred_x0(model, model_output, timestep, sample, eta=0, variance_noise=None):
    a_t = Schedule(model.scheduler).ab[int(timestep)]
    return (sample - float((1 - a_t) ** 0.5) * model_output) / float(a_t ** 0.5)


def reverse_step(model, model_output, timestep, sample, eta=0, variance_noise=None, return_pred_x0=False,
                 return_mu=False, is_ddim_inversion=False):
    S = Schedule(model.scheduler)
    t = int(timestep)
    a_t, a_p, var = S.ab[t], S.ab_prev(t), S.variance(t)
    x0 = (sample - float((1 - a_t) ** 0.5) * model_output) / float(a_t ** 0.5)
    if is_ddim_inversion:
        direction = float((1 - a_p) ** 0.5) * model_output
    else:
        direction = float((1 - a_p - (eta ** 2) * var) ** 0.5) * model_output
    prev = float(a_p ** 0.5) * x0 + direction
    mu = prev
    if eta > 0:
        if is_ddim_inversion:
            prev = prev + eta * variance_noise
        else:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, device=model_output.device)
            prev = prev + float(eta * var ** 0.5) * variance_noise
    if return_pred_x0:
        return prev, x0
    if return_mu:
        return prev, mu
    return prev


def compute_full_coeff(model, timestep, prev_timestep, eta, is_ddim_inversion=False):
    return Schedule(model.scheduler).full_coeff(int(timestep), int(prev_timestep), eta, is_ddim_inversion)


def slerp(val, low, high):
    """Spherical interpolation between the rows of two (B, D) tensors (reference
    inversion_utils.py:142-152)."""
    ln = low / torch.norm(low, dim=1, keepdim=True)
    hn = high / torch.norm(high, dim=1, keepdim=True)
    omega = torch.acos((ln * hn).sum(1))
    so = torch.sin(omega)
    return (torch.sin((1.0 - val) * omega) / so).unsqueeze(1) * low + (torch.sin(val * omega) / so).unsqueeze(1) * high


def slerp_tensor(val, low, high):
    """slerp on flattened per-item tensors, reshaped back (reference inversion_utils.py:154-160)."""
    return slerp(val, low.flatten(1), high.flatten(1)).reshape(low.shape)
