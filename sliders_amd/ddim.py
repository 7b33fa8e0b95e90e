"""Host-side DDIM scheduler state for the hot path (timestep tables and per-step fp32 coefficients).

Mirrors the scheduler the reference builds at trainscripts/textsliders/model_util.py:237-246
(diffusers DDIMScheduler: scaled_linear betas 0.00085..0.012, 1000 train steps, clip_sample False, epsilon or
v prediction (model_util.py:126), eta 0, set_alpha_to_one, leading spacing).  Only scalars live here; the tensor update
x_t -> x_{t-1} is the fused slh_cfg_ddim kernel.
"""
from __future__ import annotations

from typing import List, Tuple

import torch


class DDIMSchedule:
    fused = True        # the step is slh_cfg_ddim; sliders_amd/schedulers.py holds the tensor-op schedulers (fused = False)

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 prediction_type: str = "epsilon"):
        if prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"prediction_type {prediction_type!r}: the fused step implements epsilon and v_prediction")
        self.prediction_type = prediction_type
        self.num_train_timesteps = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = num_train_timesteps
        self.timesteps = self.make_timesteps(num_train_timesteps)

    def make_timesteps(self, n: int) -> List[int]:
        ratio = self.num_train_timesteps // n
        return [i * ratio for i in range(n)][::-1]

    def set_timesteps(self, n: int, device=None):
        self.num_inference_steps = n
        self.timesteps = self.make_timesteps(n)

    def step_coefficients(self, t: int, n_steps: int) -> Tuple[float, float, float, float]:
        """(sqrt(1-a_t), 1/sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)), each computed in fp32 like the 0-dim
        tensors diffusers multiplies into the latents."""
        prev_t = t - self.num_train_timesteps // n_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float((1 - a_t) ** 0.5), float(torch.tensor(1.0) / (a_t ** 0.5)), float(a_p ** 0.5),
                float((1 - a_p) ** 0.5))

    def step_fields(self, t: int, n_steps: int) -> dict:
        """The scheduler fields of slh_cfg_ddim_desc for timestep t (do_step = 1)."""
        cb, cia, cp, cd = self.step_coefficients(t, n_steps)
        f = dict(c_sqrt_beta_t=cb, c_inv_sqrt_alpha_t=cia, c_sqrt_alpha_prev=cp, c_dir=cd, do_step=1, v_prediction=0)
        if self.prediction_type == "v_prediction":
            f.update(v_prediction=1, c_sqrt_alpha_t=float(self.alphas_cumprod[t] ** 0.5))
        return f

    def add_noise_coefficients(self, t: int) -> Tuple[float, float]:
        a = self.alphas_cumprod[t]
        return float(a ** 0.5), float((1 - a) ** 0.5)
