"""Step helpers with the reference's names and signatures (trainscripts/textsliders/train_util.py), driving the
MI355X UNetEngine instead of a diffusers UNet.  They keep the reference's call shapes so a training script
written against the reference runs unchanged; the fused fast path is sliders_amd/trainer.py.

  get_random_noise / get_initial_latents      train_util.py:20-57   (noise drawn on the CPU, like the reference)
  concat_embeddings                           train_util.py:136-141
  predict_noise / predict_noise_xl            train_util.py:145-171 / 220-260  (CFG pair, guidance combine)
  diffusion / diffusion_xl                    train_util.py:175-196 / 263-294  (partial DDIM denoise)
  get_add_time_ids                            train_util.py:298-333
  get_optimizer / get_lr_scheduler            train_util.py:336-404
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib
from .ddim import DDIMSchedule

UNET_IN_CHANNELS = 4
VAE_SCALE_FACTOR = 8
UNET_ATTENTION_TIME_EMBED_DIM = 256
TEXT_ENCODER_2_PROJECTION_DIM = 1280
UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM = 2816


def get_random_noise(batch_size: int, height: int, width: int, generator: torch.Generator = None) -> torch.Tensor:
    return torch.randn((batch_size, UNET_IN_CHANNELS, height // VAE_SCALE_FACTOR, width // VAE_SCALE_FACTOR),
                       generator=generator, device="cpu")


def get_initial_latents(scheduler, n_imgs: int, height: int, width: int, n_prompts: int, generator=None) -> torch.Tensor:
    noise = get_random_noise(n_imgs, height, width, generator=generator).repeat(n_prompts, 1, 1, 1)
    return noise * scheduler.init_noise_sigma


def concat_embeddings(unconditional: torch.Tensor, conditional: torch.Tensor, n_imgs: int):
    return torch.cat([unconditional, conditional]).repeat_interleave(n_imgs, dim=0)


class DDIMScheduler(DDIMSchedule):
    """Tensor-level scheduler object for the reference-shaped helpers: `.timesteps`, `.set_timesteps`,
    `.scale_model_input` (identity), `.step(...).prev_sample` (slh_cfg_ddim with guidance folded out)."""

    class _Out:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    def set_timesteps(self, n: int, device=None):
        super().set_timesteps(n, device)
        self.timesteps = torch.tensor(self.timesteps, dtype=torch.int64)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.timesteps = torch.tensor(self.timesteps, dtype=torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        """x_{t-1} from epsilon with the reference's bf16 rounding points (eta = 0)."""
        eps = torch.cat([model_output, model_output]).to(torch.bfloat16).contiguous()   # guidance 1: u + 1*(t-u) = t
        x = sample.to(torch.bfloat16).contiguous()
        out = torch.empty_like(x)
        nb = x.shape[0]
        d = lib.CfgDdimDesc(eps=eps.data_ptr(), x=x.data_ptr(), out=out.data_ptr(), nb=nb, chw=x[0].numel(),
                            guidance=1.0, **self.step_fields(int(timestep), self.num_inference_steps))
        lib.call(lib.OP_CFG_DDIM, d, torch.cuda.current_stream().cuda_stream)
        return DDIMScheduler._Out(out)


def _cfg_combine(noise_pred: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
    return noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)


def predict_noise(unet, scheduler, timestep, latents, text_embeddings, guidance_scale=7.5):
    latent_model_input = torch.cat([latents] * 2)
    latent_model_input = scheduler.scale_model_input(latent_model_input, timestep)
    noise_pred = unet(latent_model_input, timestep, encoder_hidden_states=text_embeddings).sample
    return _cfg_combine(noise_pred, guidance_scale)


@torch.no_grad()
def diffusion(unet, scheduler, latents, text_embeddings, total_timesteps: int = 1000, start_timesteps=0, **kwargs):
    for timestep in scheduler.timesteps[start_timesteps:total_timesteps]:
        noise_pred = predict_noise(unet, scheduler, timestep, latents, text_embeddings, **kwargs)
        latents = scheduler.step(noise_pred, timestep, latents).prev_sample
    return latents


def predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings, add_time_ids,
                     guidance_scale=7.5, guidance_rescale=0.7):
    latent_model_input = torch.cat([latents] * 2)
    latent_model_input = scheduler.scale_model_input(latent_model_input, timestep)
    added_cond_kwargs = {"text_embeds": add_text_embeddings, "time_ids": add_time_ids}
    noise_pred = unet(latent_model_input, timestep, encoder_hidden_states=text_embeddings,
                      added_cond_kwargs=added_cond_kwargs).sample
    # the reference also evaluates rescale_noise_cfg here and discards the result (train_util.py:256-260)
    return _cfg_combine(noise_pred, guidance_scale)


@torch.no_grad()
def diffusion_xl(unet, scheduler, latents, text_embeddings, add_text_embeddings, add_time_ids,
                 guidance_scale: float = 1.0, total_timesteps: int = 1000, start_timesteps=0):
    for timestep in scheduler.timesteps[start_timesteps:total_timesteps]:
        noise_pred = predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings,
                                      add_time_ids, guidance_scale=guidance_scale, guidance_rescale=0.7)
        latents = scheduler.step(noise_pred, timestep, latents).prev_sample
    return latents


def get_add_time_ids(height: int, width: int, dynamic_crops: bool = False, dtype: torch.dtype = torch.float32):
    if dynamic_crops:
        random_scale = torch.rand(1).item() * 2 + 1
        original_size = (int(height * random_scale), int(width * random_scale))
        crops_coords_top_left = (torch.randint(0, original_size[0] - height, (1,)).item(),
                                 torch.randint(0, original_size[1] - width, (1,)).item())
        target_size = (height, width)
    else:
        original_size = (height, width)
        crops_coords_top_left = (0, 0)
        target_size = (height, width)
    add_time_ids = list(original_size + crops_coords_top_left + target_size)
    passed = UNET_ATTENTION_TIME_EMBED_DIM * len(add_time_ids) + TEXT_ENCODER_2_PROJECTION_DIM
    if passed != UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM:
        raise ValueError(f"Model expects an added time embedding vector of length "
                         f"{UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM}, but a vector of {passed} was created.")
    return torch.tensor([add_time_ids], dtype=dtype)


def get_optimizer(name: str):
    name = name.lower()
    if name == "lion":
        from .optim import Lion        # lion_pytorch.Lion's interface over the fused slh_lion kernel
        return Lion
    if name == "prodigy":
        from .optim import Prodigy     # prodigyopt.Prodigy's interface, tensor ops on the parameter's device
        return Prodigy
    if name.startswith("dadapt"):      # dadaptation 3.1's two classes (train_util.py:339-346), same interface, tensor ops
        from .optim import DAdaptAdam, DAdaptLion
        if name == "dadaptadam":
            return DAdaptAdam
        if name == "dadaptlion":
            return DAdaptLion
        raise ValueError("DAdapt optimizer must be dadaptadam or dadaptlion")
    if name.endswith("8bit"):
        raise ValueError(f"optimizer {name} needs a package that is not installed in this image (bitsandbytes)")
    if name == "adam":
        return torch.optim.Adam
    if name == "adamw":
        return torch.optim.AdamW
    raise ValueError("Optimizer must be adam, adamw, lion or Prodigy")


def get_lr_scheduler(name: Optional[str], optimizer, max_iterations: Optional[int], lr_min: Optional[float] = None,
                     **kwargs):
    if name == "cosine":
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=max_iterations, eta_min=lr_min, **kwargs)
    if name == "cosine_with_restarts":
        return torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer, T_0=max_iterations // 10, T_mult=2,
                                                                    eta_min=lr_min, **kwargs)
    if name == "step":
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=max_iterations // 100, gamma=0.999, **kwargs)
    if name == "constant":
        return torch.optim.lr_scheduler.ConstantLR(optimizer, factor=1, **kwargs)
    if name == "linear":
        return torch.optim.lr_scheduler.LinearLR(optimizer, factor=0.5, total_iters=max_iterations // 100, **kwargs)
    raise ValueError("Scheduler must be cosine, cosine_with_restarts, step, linear or constant")


def get_random_resolution_in_bucket(bucket_resolution: int = 512):
    max_resolution = bucket_resolution
    min_resolution = bucket_resolution // 2
    step = 64
    min_step = min_resolution // step
    max_step = max_resolution // step
    height = torch.randint(min_step, max_step, (1,)).item() * step
    width = torch.randint(min_step, max_step, (1,)).item() * step
    return height, width
