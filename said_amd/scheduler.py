"""Host side of the DDIM noise scheduler.

Mirrors the slice of ``diffusers==0.19.*`` ``DDIMScheduler`` that the reference
touches (/root/reference/said/model/diffusion.py:100-104, 179, 247, 271-272,
361, 370, 378, 404, 413, 424-426, 441-443, 451-454): same constructor
arguments, ``config``, ``timesteps``, ``alphas_cumprod``, ``init_noise_sigma``,
``set_timesteps``, ``scale_model_input``, ``step``, ``add_noise``,
``get_velocity``.  Only the *tables* are computed here (fp32 torch ops on the
CPU, in that library's op order); every elementwise tensor update runs in the
HIP engine (``said_ddim_step`` / ``said_axpby`` / the fused loop).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional, Union

import numpy as np
import torch

from . import _engine

NCOEF = _engine.NCOEF


def _betas_squaredcos_cap_v2(n: int, max_beta: float = 0.999) -> torch.Tensor:
    def alpha_bar(t: float) -> float:
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    return torch.tensor([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)], dtype=torch.float32)


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class DDIMScheduler:
    """DDIM scheduler with diffusers-0.19 defaults: clip_sample=True (range 1.0),
    set_alpha_to_one=True, steps_offset=0, timestep_spacing="leading"."""

    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2",
                 prediction_type: str = "epsilon", **kwargs):
        if beta_schedule != "squaredcos_cap_v2":
            raise NotImplementedError(f"beta_schedule={beta_schedule!r}: the SAiD path only uses 'squaredcos_cap_v2'")
        if prediction_type not in _engine.PRED:
            raise ValueError(f"prediction_type must be one of {list(_engine.PRED)}")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_schedule=beta_schedule,
                                      prediction_type=prediction_type, clip_sample=True, clip_sample_range=1.0,
                                      set_alpha_to_one=True, steps_offset=0, timestep_spacing="leading", **kwargs)
        self.betas = _betas_squaredcos_cap_v2(num_train_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._engine: Optional[_engine.Engine] = None  # attached by the owning SAID model

    # ---- tables -------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None) -> None:
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("`num_inference_steps` cannot be larger than `num_train_timesteps`")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def _coef_row(self, timestep: int, eta: float, next_timestep: Optional[int]) -> np.ndarray:
        """One row of the engine's coefficient table (include/said_hip.h SAID_COEF_*),
        each entry a 0-dim fp32 tensor op in DDIMScheduler.step's order."""
        prev_t = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_p = 1 - a_p
        variance = (b_p / b_t) * (1 - a_t / a_p)
        std_dev_t = eta * variance ** (0.5)
        row = np.zeros(NCOEF, dtype=np.float32)
        row[0] = float(a_t ** (0.5))
        row[1] = float(b_t ** (0.5))
        row[2] = float(a_p ** (0.5))
        row[3] = float((1 - a_p - std_dev_t ** 2) ** (0.5))
        row[4] = float(std_dev_t)
        if next_timestep is None:
            row[5], row[6] = 1.0, 0.0
        else:
            a_n = self.alphas_cumprod[next_timestep]
            row[5] = float(a_n ** 0.5)
            row[6] = float((1 - a_n) ** 0.5)
        return row

    def coef_table(self, timesteps: np.ndarray, eta: float) -> np.ndarray:
        """Rows for consecutive loop steps; the mask-blend columns use the *next*
        loop timestep (diffusion.py:449-454), identity on the last step.

        One batched fp32 expression over all steps: every element goes through the same fp32 ops in the same
        order as ``_coef_row`` (0-dim tensor ops, DDIMScheduler.step's order), so the table is bit-identical to the
        row-wise form (tests/test_host_cpu.py asserts it) — and a 1000-step table costs ~0.2 ms of host time
        instead of ~25 ms, which a single-clip caller would otherwise pay before the first kernel."""
        n = len(timesteps)
        if n == 0:
            return np.zeros((0, NCOEF), np.float32)
        ts = torch.as_tensor(np.asarray(timesteps, dtype=np.int64))
        ac = self.alphas_cumprod
        prev = ts - self.config.num_train_timesteps // self.num_inference_steps
        a_t = ac[ts]
        a_p = torch.where(prev >= 0, ac[prev.clamp(min=0)], self.final_alpha_cumprod.to(ac.dtype))
        b_t = 1 - a_t
        b_p = 1 - a_p
        variance = (b_p / b_t) * (1 - a_t / a_p)
        std_dev_t = eta * variance ** (0.5)
        tab = torch.zeros(n, NCOEF, dtype=torch.float32)
        tab[:, 0] = a_t ** (0.5)
        tab[:, 1] = b_t ** (0.5)
        tab[:, 2] = a_p ** (0.5)
        tab[:, 3] = (1 - a_p - std_dev_t ** 2) ** (0.5)
        tab[:, 4] = std_dev_t
        tab[:, 5], tab[:, 6] = 1.0, 0.0
        if n > 1:
            a_n = ac[ts[1:]]
            tab[:-1, 5] = a_n ** 0.5
            tab[:-1, 6] = (1 - a_n) ** 0.5
        return tab.numpy()

    def coef_table_rowwise(self, timesteps: np.ndarray, eta: float) -> np.ndarray:
        """The same table built row by row with 0-dim tensor ops (reference form; used by the tests)."""
        n = len(timesteps)
        return np.stack([self._coef_row(int(timesteps[k]), eta, int(timesteps[k + 1]) if k + 1 < n else None)
                         for k in range(n)]) if n else np.zeros((0, NCOEF), np.float32)

    # ---- tensor updates (HIP engine) ------------------------------------------
    def _need_engine(self) -> _engine.Engine:
        if self._engine is None:
            raise _engine.EngineError("scheduler is not attached to a HIP engine (construct it through SAID_UNet1D on an MI355X)")
        return self._engine

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise: Optional[torch.Tensor] = None,
             return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output=True is not used by the SAiD path")
        row = self._coef_row(int(timestep), float(eta), None)
        noise = None
        if eta > 0:
            noise = variance_noise if variance_noise is not None else torch.randn(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
        prev = self._need_engine().ddim_step(model_output, sample, row, self.config.prediction_type, step_noise=noise)
        return DDIMSchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def _sqrt_pair(self, timesteps: torch.Tensor):
        ts = torch.as_tensor(timesteps).to("cpu").reshape(-1)
        ac = self.alphas_cumprod
        sa = (ac[ts] ** 0.5).flatten()
        sb = ((1 - ac[ts]) ** 0.5).flatten()
        return sa.tolist(), sb.tolist()

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        sa, sb = self._sqrt_pair(timesteps)
        B = original_samples.shape[0]
        if len(sa) == 1 and B > 1:
            sa, sb = sa * B, sb * B
        return self._need_engine().axpby(sa, original_samples, sb, noise)

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        sa, sb = self._sqrt_pair(timesteps)
        B = sample.shape[0]
        if len(sa) == 1 and B > 1:
            sa, sb = sa * B, sb * B
        return self._need_engine().axpby(sa, noise, [-v for v in sb], sample)
