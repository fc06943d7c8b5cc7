"""DDIM scheduler restatement — PARITY UNPINNED (see oracle/__init__.py).

Restates ``diffusers==0.19.*`` ``DDIMScheduler`` as the reference configures it
(/root/reference/said/model/diffusion.py:100-104:
``num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", prediction_type``)
with that version's defaults: ``clip_sample=True``, ``clip_sample_range=1.0``,
``set_alpha_to_one=True``, ``steps_offset=0``, ``timestep_spacing="leading"``,
``thresholding=False``, ``use_clipped_model_output=False``.

Arithmetic is written as separate fp32 tensor ops in the library's order so the
HIP epilogue can be compared bit-for-bit given identical inputs.
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch


def betas_for_alpha_bar(num_diffusion_timesteps: int, max_beta: float = 0.999) -> torch.Tensor:
    """``squaredcos_cap_v2``: Python float64 per-step ratio, stored as fp32."""

    def alpha_bar(t: float) -> float:
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    betas: List[float] = []
    for i in range(num_diffusion_timesteps):
        t1 = i / num_diffusion_timesteps
        t2 = (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return torch.tensor(betas, dtype=torch.float32)


class OracleDDIM:
    """Minimal DDIMScheduler surface used by SAID (call sites diffusion.py:179,
    247, 271-272, 361, 370, 378, 404, 413, 424-426, 441-443, 451-454)."""

    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, prediction_type: str = "epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.betas = betas_for_alpha_bar(num_train_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def _get_variance(self, timestep: int, prev_timestep: int) -> torch.Tensor:
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_p = 1 - a_p
        return (b_p / b_t) * (1 - a_t / a_p)

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0,
             variance_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns ``prev_sample``.  ``variance_noise`` must be supplied when
        eta > 0 (the caller owns the RNG so CPU and GPU see identical noise)."""
        timestep = int(timestep)
        prev_timestep = timestep - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** (0.5) * model_output) / a_t ** (0.5)
            eps = model_output
        elif self.prediction_type == "sample":
            x0 = model_output
            eps = (sample - a_t ** (0.5) * x0) / b_t ** (0.5)
        elif self.prediction_type == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise ValueError(self.prediction_type)
        x0 = x0.clamp(-1.0, 1.0)  # clip_sample=True, clip_sample_range=1.0
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** (0.5)
        direction = (1 - a_p - std_dev_t ** 2) ** (0.5) * eps
        prev_sample = a_p ** (0.5) * x0 + direction
        if eta > 0:
            if variance_noise is None:
                raise ValueError("eta > 0 needs variance_noise")
            prev_sample = prev_sample + std_dev_t * variance_noise
        return prev_sample

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        ac = self.alphas_cumprod.to(dtype=original.dtype)
        timesteps = torch.as_tensor(timesteps)
        sa = (ac[timesteps] ** 0.5).flatten()
        while sa.dim() < original.dim():
            sa = sa.unsqueeze(-1)
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sb.dim() < original.dim():
            sb = sb.unsqueeze(-1)
        return sa * original + sb * noise

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        ac = self.alphas_cumprod.to(dtype=sample.dtype)
        timesteps = torch.as_tensor(timesteps)
        sa = (ac[timesteps] ** 0.5).flatten()
        while sa.dim() < sample.dim():
            sa = sa.unsqueeze(-1)
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sb.dim() < sample.dim():
            sb = sb.unsqueeze(-1)
        return sa * noise - sb * sample


def rescale_noise_cfg(noise_cfg: torch.Tensor, noise_pred_text: torch.Tensor, guidance_rescale: float) -> torch.Tensor:
    """diffusers ``rescale_noise_cfg`` (call site diffusion.py:436-439)."""
    std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg
