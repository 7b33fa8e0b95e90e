"""The reference's non-default noise schedulers: "ddpm", "lms", "euler_a" (trainscripts/textsliders/model_util.py:247-274).

The default scheduler (DDIM) is the fused HIP step `slh_cfg_ddim` (sliders_amd/ddim.py).  These three are selected by
`train.noise_scheduler` and do per step, next to a ~28 ms UNet pass, a handful of scalar-times-tensor operations on one
(bs, 4, H, W) latent, two of them with fresh device noise.  They are written as the SAME tensor-op sequence the library the
reference calls executes (diffusers 0.20.2 `*.step`, `scale_model_input`, `set_timesteps`; consumed at
train_util.py:55/156/193), on whatever device the latents live on, in the latents' dtype, with fp32 0-dim scalars - so
the rounding points are the reference's by construction.  No HIP kernel is involved and none is needed (launch-bound
elementwise work, < 0.3 % of a denoise step); the UNet pass and the guidance combine stay the HIP path, and nothing here
touches the CPU when the latents are device tensors (the scalar tables are host-side like DDIMSchedule's).

Interface = what the reference's loop uses: `.timesteps`, `.set_timesteps(n, device)`, `.init_noise_sigma`,
`.scale_model_input(sample, timestep)`, `.step(model_output, timestep, sample).prev_sample`.  `fused = False` tells
SliderTrainer to run its scheduler-agnostic denoise loop.  Extension for tests: `step(..., noise=)` supplies the
per-step noise instead of drawing it.

CPU-validated against oracle/sched_oracle.py (tests/test_schedulers.py); the trainer branch that drives them was written
without GPU access at the end of round 2 and has not run on hardware yet.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

_PRED = ("epsilon", "v_prediction")


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class _Base:
    fused = False

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 prediction_type: str = "epsilon"):
        if prediction_type not in _PRED:
            raise ValueError(f"prediction_type {prediction_type!r}: epsilon and v_prediction are implemented")
        self.prediction_type = prediction_type
        self.num_train_timesteps = num_train_timesteps
        # scaled_linear, fp32 like the library
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_inference_steps = None

    def _randn(self, like: torch.Tensor, generator) -> torch.Tensor:
        return torch.randn(like.shape, generator=generator, device=like.device, dtype=like.dtype)


class DDPMScheduler(_Base):
    """model_util.py:247-256: DDPMScheduler(scaled_linear 0.00085..0.012, 1000 steps, clip_sample=False); defaults
    variance_type "fixed_small", timestep_spacing "leading"."""
    init_noise_sigma = 1.0

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.one = torch.tensor(1.0)
        self.timesteps = torch.arange(self.num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, n: int, device=None):
        if n > self.num_train_timesteps:
            raise ValueError(f"num_inference_steps {n} > num_train_timesteps {self.num_train_timesteps}")
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def _variance(self, t: int, prev_t: int) -> torch.Tensor:
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_p
        return torch.clamp((1 - a_p) / (1 - a_t) * cur_beta, min=1e-20)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None,
             noise: Optional[torch.Tensor] = None) -> SchedulerOutput:
        t = int(timestep)
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        prev_t = t - self.num_train_timesteps // n
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        if self.prediction_type == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        else:
            pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
        pred_original_sample_coeff = (alpha_prod_t_prev ** 0.5 * current_beta_t) / beta_prod_t
        current_sample_coeff = current_alpha_t ** 0.5 * beta_prod_t_prev / beta_prod_t
        pred_prev_sample = pred_original_sample_coeff * pred_original_sample + current_sample_coeff * sample
        if t > 0:
            variance_noise = noise if noise is not None else self._randn(model_output, generator)
            pred_prev_sample = pred_prev_sample + (self._variance(t, prev_t) ** 0.5) * variance_noise
        return SchedulerOutput(pred_prev_sample, pred_original_sample)


class _SigmaBase(_Base):
    """sigma-space schedulers: float timesteps on a linspace grid, sigmas interpolated from the training sigmas, model
    input divided by sqrt(sigma^2 + 1); timestep_spacing "linspace" (the library default), so init_noise_sigma is the
    largest sigma."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.set_timesteps(self.num_train_timesteps)
        self.num_inference_steps = None

    timestep_spacing = "linspace"
    steps_offset = 0

    def set_timesteps(self, n: int, device=None):
        self.num_inference_steps = n
        if self.timestep_spacing == "linspace":
            timesteps = np.linspace(0, self.num_train_timesteps - 1, n, dtype=float)[::-1].copy()
        else:       # "leading" (+ steps_offset): what the SDXL checkpoints' scheduler_config.json selects
            ratio = self.num_train_timesteps // n
            timesteps = (np.arange(0, n) * ratio).round()[::-1].copy().astype(float) + self.steps_offset
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
        # host-side tables: a 0-dim fp32 CPU tensor multiplies a device tensor like the library's 0-dim device scalar
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps)
        self._on_set_timesteps()

    def _on_set_timesteps(self):
        pass

    @property
    def init_noise_sigma(self):
        if self.timestep_spacing in ("linspace", "trailing"):
            return self.sigmas.max()
        return (self.sigmas.max() ** 2 + 1) ** 0.5

    def index_of(self, timestep) -> int:
        hit = (self.timesteps == float(timestep)).nonzero()
        if hit.numel() != 1:
            raise ValueError(f"timestep {float(timestep)} is not one of the {len(self.timesteps)} scheduler timesteps")
        return int(hit.item())

    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        sigma = self.sigmas[self.index_of(timestep)]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def _pred_original(self, model_output, sample, sigma):
        if self.prediction_type == "epsilon":
            return sample - sigma * model_output
        return model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + (sample / (sigma ** 2 + 1))


class EulerAncestralDiscreteScheduler(_SigmaBase):
    """model_util.py:266-274"""

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None,
             noise: Optional[torch.Tensor] = None) -> SchedulerOutput:
        step_index = self.index_of(timestep)
        sigma = self.sigmas[step_index]
        pred_original_sample = self._pred_original(model_output, sample, sigma)
        sigma_from = self.sigmas[step_index]
        sigma_to = self.sigmas[step_index + 1]
        sigma_up = (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        derivative = (sample - pred_original_sample) / sigma
        dt = sigma_down - sigma
        prev_sample = sample + derivative * dt
        if noise is None:
            noise = self._randn(model_output, generator)
        prev_sample = prev_sample + noise * sigma_up
        return SchedulerOutput(prev_sample, pred_original_sample)


class EulerDiscreteScheduler(_SigmaBase):
    """The scheduler the SDXL pipelines of generate_images_xl.py / XL-sliders-inference.ipynb run with (it comes from the
    checkpoint's scheduler_config.json: EulerDiscreteScheduler, timestep_spacing "leading", steps_offset 1); s_churn = 0,
    i.e. the deterministic Euler step.  Not one of train.noise_scheduler's names - inference only (SliderSampler)."""

    def __init__(self, *a, timestep_spacing: str = "leading", steps_offset: int = 1, **k):
        if timestep_spacing not in ("linspace", "leading"):
            raise ValueError(f"timestep_spacing {timestep_spacing!r}: linspace and leading are implemented")
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        super().__init__(*a, **k)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None, noise=None) -> SchedulerOutput:
        step_index = self.index_of(timestep)
        sigma = self.sigmas[step_index]
        sigma_hat = sigma                      # gamma = 0 (s_churn = 0): no noise is mixed in
        pred_original_sample = self._pred_original(model_output, sample, sigma_hat)
        derivative = (sample - pred_original_sample) / sigma_hat
        dt = self.sigmas[step_index + 1] - sigma_hat
        return SchedulerOutput(sample + derivative * dt, pred_original_sample)


class LMSDiscreteScheduler(_SigmaBase):
    """model_util.py:257-265; linear multistep of order <= 4 over the derivative history (cleared by set_timesteps,
    which the reference calls at the start of every iteration, train_lora_xl.py:164)."""

    def _on_set_timesteps(self):
        self.derivatives: List[torch.Tensor] = []
        self._sigma_list = self.sigmas.tolist()

    def get_lms_coefficient(self, order: int, t: int, current_order: int) -> float:
        from scipy import integrate

        sig = self._sigma_list      # the fp32 table as python floats (the library indexes its fp32 tensor here)

        def lms_derivative(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - sig[t - k]) / (sig[t - current_order] - sig[t - k])
            return prod

        return integrate.quad(lms_derivative, sig[t], sig[t + 1], epsrel=1e-4)[0]

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, order: int = 4, generator=None,
             noise=None) -> SchedulerOutput:
        step_index = self.index_of(timestep)
        sigma = self.sigmas[step_index]
        pred_original_sample = self._pred_original(model_output, sample, sigma)
        derivative = (sample - pred_original_sample) / sigma
        self.derivatives.append(derivative)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(step_index + 1, order)
        lms_coeffs = [self.get_lms_coefficient(order, step_index, curr_order) for curr_order in range(order)]
        prev_sample = sample + sum(coeff * derivative for coeff, derivative in zip(lms_coeffs, reversed(self.derivatives)))
        return SchedulerOutput(prev_sample, pred_original_sample)


def denoise(sched, predict, latents: torch.Tensor, k: int, num_steps: int, generator=None, device=None) -> torch.Tensor:
    """The reference's partial denoise (train_util.diffusion[_xl], train_util.py:175-196 / 263-294, entered at
    train_lora_xl.py:164-227) over the first k of num_steps timesteps: set_timesteps, then per step scale_model_input ->
    guided prediction -> step.  `predict(model_input, timestep)` returns the guided model output (UNet pass + CFG
    combine: the HIP path in SliderTrainer, a stub in the CPU tests)."""
    sched.set_timesteps(num_steps, device=device)
    for i in range(k):
        t = sched.timesteps[i]
        out = predict(sched.scale_model_input(latents, t), t)
        latents = sched.step(out, t, latents, generator=generator).prev_sample
    return latents


def create(name: str, prediction_type: str = "epsilon"):
    """the non-DDIM half of create_noise_scheduler (model_util.py:230-277)"""
    key = name.lower().replace(" ", "_")
    table = {"ddpm": DDPMScheduler, "lms": LMSDiscreteScheduler, "euler_a": EulerAncestralDiscreteScheduler}
    if key not in table:
        raise ValueError(f"Unknown scheduler name: {key}")
    return table[key](prediction_type=prediction_type)
