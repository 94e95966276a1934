"""Scheduler algebra of the h-Edit loops (oracle; see oracle/__init__.py).

Follows text-guided/inversion/inversion_utils.py of the reference:
  get_variance            :38-56
  reverse_step            :58-126
  reverse_step_pred_x0    :128-140
  compute_full_coeff      :168-195
All arithmetic is done on 0-d fp32 tensors taken from ``scheduler.alphas_cumprod`` exactly like
the reference does, so results agree to fp32 rounding.
"""
import torch


def _prev_t(sch, t):
    return t - sch.config.num_train_timesteps // sch.num_inference_steps


def _abar_prev(sch, t):
    p = _prev_t(sch, t)
    return sch.alphas_cumprod[p] if p >= 0 else sch.final_alpha_cumprod


def get_variance(sch, t):
    """sigma_t^2 = (1-abar_prev)/(1-abar_t) * (1 - abar_t/abar_prev)   (inversion_utils.py:49-55)"""
    a_t = sch.alphas_cumprod[t]
    a_p = _abar_prev(sch, t)
    return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)


def tweedie_x0(sch, eps, t, x):
    """x0_hat = (x - sqrt(1-abar_t) eps)/sqrt(abar_t)   (inversion_utils.py:134-138)"""
    a_t = sch.alphas_cumprod[t]
    return (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5


def reverse_step(sch, eps, t, x, eta=0.0, z=None, ddim_inv=False, want_x0=False):
    """One reverse step x_t -> x_{t-1} (inversion_utils.py:84-119).

    ddim_inv=True is the reference's h-Edit-D convention: direction uses sqrt(1-abar_prev) and the
    stored 'noise' z is added un-scaled (:102-103,:112-114)."""
    a_t = sch.alphas_cumprod[t]
    a_p = _abar_prev(sch, t)
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    var = get_variance(sch, t)
    if ddim_inv:
        direction = (1 - a_p) ** 0.5 * eps
    else:
        direction = (1 - a_p - (eta ** 2) * var) ** 0.5 * eps
    prev = a_p ** 0.5 * x0 + direction
    if eta > 0:
        if ddim_inv:
            prev = prev + eta * z
        else:
            if z is None:
                z = torch.randn(eps.shape)
            prev = prev + eta * var ** 0.5 * z
    return (prev, x0) if want_x0 else prev


def full_coeff(sch, t, t_prev, eta, ddim_inv=False):
    """sqrt(1 - abar_{t_prev} - omega^2)   (inversion_utils.py:183-195)."""
    ab = sch.alphas_cumprod
    sig = (1 - ab) ** 0.5
    a = ab ** 0.5
    omega = eta * (sig[t_prev] / (sig[t] * a[t_prev])) * ((ab[t_prev] - ab[t]) ** 0.5)
    if ddim_inv:
        omega = 0
    return (1 - ab[t_prev] - omega ** 2) ** 0.5


def edit_coeff(sch, t, t_prev, eta, ddim_inv=False):
    """The multiplier of the correction term f(.) in the h-Edit update:
    full_coeff - sqrt(1-abar_t) * sqrt(abar_{t_prev}/abar_t)
    (text-guided/inversion/p2p_h_edit.py:664-665, twins :347-348, :508-509, :139-140)."""
    ab = sch.alphas_cumprod
    ratio = (ab ** 0.5)[t_prev] / (ab ** 0.5)[t]
    return full_coeff(sch, t, t_prev, eta, ddim_inv) - ((1 - ab) ** 0.5)[t] * ratio
