"""CPU oracle: DDIM scheduler as the reference configures it.

TEST INFRASTRUCTURE ONLY (see unet_oracle.py header).

Restates diffusers-0.20.2 `DDIMScheduler` (third-party, not vendored, not installed) for the
configuration built at trainscripts/textsliders/model_util.py:237-246:
    DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                  num_train_timesteps=1000, clip_sample=False, prediction_type="epsilon")
with the library defaults set_alpha_to_one=True, steps_offset=0, timestep_spacing="leading",
eta=0.  PARITY UNPINNED against the real package (cannot be imported
here); pinned against the closed-form DDIM update in tests/test_oracle.py.

Call sites in the reference: `scheduler.set_timesteps` (train_lora_xl.py:164-166, 229),
`scheduler.timesteps[...]` (train_lora_xl.py:231-233, train_util.py:277),
`scheduler.scale_model_input` (train_util.py:156, 234), `scheduler.step(...).prev_sample`
(train_util.py:193, 291), `scheduler.init_noise_sigma` (train_util.py:55).
"""
from __future__ import annotations

import torch


class _StepOutput:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                 beta_end: float = 0.012, prediction_type: str = "epsilon"):
        assert prediction_type in ("epsilon", "v_prediction")   # model_util.py:126
        self.prediction_type = prediction_type
        self.num_train_timesteps = num_train_timesteps
        # "scaled_linear": linspace in sqrt space, fp32
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)  # set_alpha_to_one=True
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        # "leading" spacing, steps_offset = 0
        ts = (torch.arange(0, num_inference_steps, dtype=torch.float64) * step_ratio).round()
        self.timesteps = ts.flip(0).to(torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample  # identity for DDIM

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * (1 - alpha_prod_t / alpha_prod_t_prev)
        std_dev_t = eta * variance ** 0.5
        # diffusers writes (epsilon prediction, no clipping):
        #   pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        #   pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * model_output
        #   prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        # where the coefficients are 0-dim fp32 CPU tensors and the operands bf16 tensors ON THE GPU (the
        # reference trains on cuda, train_lora_xl.py:414).  torch's CUDA binary kernels keep a CPU scalar in
        # fp32 opmath (no rounding of the coefficient to bf16) and divide by a CPU scalar as a * (1 / b);
        # every op rounds its result once to the tensor dtype.  _smul restates exactly that, so this oracle
        # gives the same bits on CPU as the reference's ops give on a GPU.
        dt = sample.dtype

        def _smul(scalar, tensor):
            return (tensor.float() * scalar.float()).to(dt)

        c_sqrt_beta = beta_prod_t ** 0.5
        c_inv_sqrt_alpha = torch.tensor(1.0) / (alpha_prod_t ** 0.5)
        c_dir = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5
        c_sqrt_alpha_prev = alpha_prod_t_prev ** 0.5
        if self.prediction_type == "v_prediction":
            # diffusers:  pred_original_sample = (alpha_prod_t**0.5) * sample - (beta_prod_t**0.5) * model_output
            #             pred_epsilon = (alpha_prod_t**0.5) * model_output + (beta_prod_t**0.5) * sample
            c_sqrt_alpha = alpha_prod_t ** 0.5
            pred_original_sample = _smul(c_sqrt_alpha, sample) - _smul(c_sqrt_beta, model_output)
            pred_epsilon = _smul(c_sqrt_alpha, model_output) + _smul(c_sqrt_beta, sample)
            prev_sample = _smul(c_sqrt_alpha_prev, pred_original_sample) + _smul(c_dir, pred_epsilon)
            return _StepOutput(prev_sample, pred_original_sample)
        pred_original_sample = _smul(c_inv_sqrt_alpha, sample - _smul(c_sqrt_beta, model_output))
        pred_sample_direction = _smul(c_dir, model_output)
        prev_sample = _smul(c_sqrt_alpha_prev, pred_original_sample) + pred_sample_direction
        return _StepOutput(prev_sample, pred_original_sample)

    def add_noise(self, original_samples, noise, timesteps):
        """imagesliders/train_util.py:231-233 path (image sliders)."""
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        a = ac[timesteps] ** 0.5
        s = (1 - ac[timesteps]) ** 0.5
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
            s = s.unsqueeze(-1)
        return a * original_samples + s * noise

    def step_coefficients(self, timestep):
        """(sqrt(1-a_t), 1/sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)) as python floats of fp32 values."""
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float((1 - a_t) ** 0.5), float(torch.tensor(1.0) / (a_t ** 0.5)), float(a_p ** 0.5),
                float((1 - a_p - (0.0 * a_p) ** 2) ** 0.5))
