"""Noise schedule of the diffusion process (host side, float64 -> float32 once at construction).

Mirrors what GaussianDiffusion.__init__ registers (reference src/models/modules/cond_DDPM.py:271-287
for the beta schedules, :336-377 for the 13 buffers); the HIP path consumes five of them through
cddpm_set_schedule and two through cddpm_q_sample.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    "p2_loss_weight",
)


def linear_beta_schedule(timesteps: int) -> torch.Tensor:
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    """https://openreview.net/forum?id=-NEXDKk8gZ ; betas clipped to [0, 0.999]"""
    grid = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    abar = torch.cos(((grid / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0, 0.999)


def schedule_buffers(timesteps: int, beta_schedule: str = "cosine", p2_loss_weight_gamma: float = 0.0,
                     p2_loss_weight_k: float = 1.0) -> Dict[str, torch.Tensor]:
    if beta_schedule == "linear":
        betas = linear_beta_schedule(timesteps)
    elif beta_schedule == "cosine":
        betas = cosine_beta_schedule(timesteps)
    else:
        raise ValueError(f"unknown beta schedule {beta_schedule}")
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = F.pad(abar[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - abar_prev) / (1.0 - abar)
    vals = {
        "betas": betas,
        "alphas_cumprod": abar,
        "alphas_cumprod_prev": abar_prev,
        "sqrt_alphas_cumprod": abar.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - abar).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - abar).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / abar).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / abar - 1).sqrt(),
        "posterior_variance": post_var,
        # log clipped: the posterior variance is 0 at the start of the chain
        "posterior_log_variance_clipped": post_var.clamp(min=1e-20).log(),
        "posterior_mean_coef1": betas * abar_prev.sqrt() / (1.0 - abar),
        "posterior_mean_coef2": (1.0 - abar_prev) * alphas.sqrt() / (1.0 - abar),
        "p2_loss_weight": (p2_loss_weight_k + abar / (1 - abar)) ** -p2_loss_weight_gamma,
    }
    return {k: vals[k].to(torch.float32) for k in BUFFER_NAMES}
