"""Slider inference on the MI355X engine: the sampling loop the reference's consumers run over a trained slider
(eval-scripts/generate_images_sd1.py:136-171, trainscripts/textsliders/generate_images_xl.py:39-394, the inference
notebooks):

    latents = randn(seed) * init_noise_sigma
    for t in scheduler.timesteps (DDIM, `ddim_steps`):
        network.set_lora_slider(scale = 0 if t > start_noise else scale)
        with network: eps = unet(cat([latents] * 2), t, cat([uncond, text]) [, added_cond_kwargs])
        eps = eps_uncond + guidance_scale * (eps_text - eps_uncond)
        latents = scheduler.step(eps, t, latents).prev_sample
    image = vae.decode(latents / scaling_factor); image = (image / 2 + 0.5).clamp(0, 1)

Every UNet evaluation is one replay of the engine's LoRA-on command buffer (the adapter scale is a device scalar, so
gating it per step costs nothing); CFG combine + DDIM step is the fused slh_cfg_ddim kernel; the VAE decoder is the fp32
command buffer of sliders_amd/vae.py.  This is what lets a slider trained here be evaluated here (CLIP-score direction,
eval-scripts/clip_score.py) without the diffusers pipelines.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib
from . import schedulers
from .ddim import DDIMSchedule
from .lora_store import LoraStore
from .unet import UNetEngine
from .vae import VaeDecoder


class SliderSampler:
    def __init__(self, engine: UNetEngine, store: Optional[LoraStore] = None, decoder: Optional[VaeDecoder] = None,
                 prediction_type: str = "epsilon", scheduler: str = "ddim", scheduler_seed: int = 0):
        """scheduler: "ddim" (fused HIP step), "lms" (what eval-scripts/generate_images_sd1.py:51 and the SD-1 notebook
        construct), "euler" (the SDXL checkpoints' scheduler_config: what generate_images_xl.py's pipeline runs), "euler_a",
        "ddpm" - the non-DDIM ones from sliders_amd/schedulers.py."""
        self.eng, self.store, self.decoder = engine, store, decoder
        if store is not None and engine.lora is not store:
            engine.attach_lora(store)
        if scheduler.lower().replace(" ", "_") == "ddim":
            self.sched = DDIMSchedule(prediction_type=prediction_type)
        elif scheduler.lower() == "euler":     # the SDXL checkpoints' own scheduler (generate_images_xl.py pipelines)
            self.sched = schedulers.EulerDiscreteScheduler(prediction_type=prediction_type)
            self.sched_generator = None
        else:
            self.sched = schedulers.create(scheduler, prediction_type)
            self.sched_generator = torch.Generator(device=engine.device)
            self.sched_generator.manual_seed(int(scheduler_seed))

    @torch.no_grad()
    def sample_latents(self, ctx: torch.Tensor, noise: torch.Tensor, scale: float = 0.0, start_noise: int = 750,
                       ddim_steps: int = 50, guidance_scale: float = 7.5, pooled: Optional[torch.Tensor] = None,
                       time_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ctx: (2*bs, 77, D) = cat([unconditional, text]); noise: (bs, 4, h, w) UNIT noise (it is multiplied by the
        scheduler's init_noise_sigma here, generate_images_sd1.py:165); returns the final latents (bs, 4, h, w) bf16.
        LoRA multiplier per step: 0 while t > start_noise, `scale` after."""
        eng = self.eng
        bs, _, h, w = noise.shape
        mode = "on" if self.store is not None else "off"
        p = eng.plan(2 * bs, h, w, mode)
        io = p.io
        io["ctx"].tensor.copy_(ctx.to(torch.bfloat16))
        if eng.cfg.is_xl:
            if time_ids is None:
                time_ids = torch.tensor([[h * 8.0, w * 8.0, 0.0, 0.0, h * 8.0, w * 8.0]] * (2 * bs))
            io["time_ids"].tensor.copy_(time_ids.to(device=eng.device, dtype=torch.float32).reshape(2 * bs, 6))
            io["add_in"].tensor[:, : eng.cfg.pooled_dim].copy_(pooled.to(torch.bfloat16))
        smp = io["sample"]
        if not self.sched.fused:
            return self._sample_unfused(p, noise, scale, start_noise, ddim_steps, guidance_scale)
        lat = noise.to(eng.device, torch.bfloat16)
        smp.tensor[:bs].copy_(lat)
        smp.tensor[bs:].copy_(lat)
        s = torch.cuda.current_stream().cuda_stream
        chw = eng.cfg.out_channels * h * w
        half = bs * chw * 2
        for i, t in enumerate(self.sched.make_timesteps(ddim_steps)):
            if self.store is not None:
                eng.set_lora(True, 0.0 if t > start_noise else float(scale))
            io["t"].tensor.fill_(float(t))
            (p.prog if (i == 0 or p.prog_text_cached is None) else p.prog_text_cached).run(s)
            d = lib.CfgDdimDesc(eps=io["eps"].ptr, x=smp.ptr, out=smp.ptr, out2=smp.ptr + half, nb=bs, chw=chw,
                                guidance=float(guidance_scale), **self.sched.step_fields(t, ddim_steps))
            lib.call(lib.OP_CFG_DDIM, d, s)
        if self.store is not None:
            eng.set_lora(False)
        return smp.tensor[:bs].clone()

    def _sample_unfused(self, p, noise, scale, start_noise, steps, guidance_scale):
        """generate_images_sd1.py:160-185 with a tensor-op scheduler: scale_model_input -> UNet replay -> guidance combine
        (slh_cfg_ddim, do_step = 0) -> scheduler.step on the device latents"""
        eng, sch, io = self.eng, self.sched, p.io
        bs = noise.shape[0]
        s = torch.cuda.current_stream().cuda_stream
        sch.set_timesteps(steps, device=eng.device)
        lat = (noise.to(eng.device, torch.float32) * sch.init_noise_sigma).to(torch.bfloat16)
        eps = torch.empty_like(lat)
        chw = lat[0].numel()
        for i, t in enumerate(sch.timesteps):
            if self.store is not None:
                eng.set_lora(True, 0.0 if float(t) > start_noise else float(scale))
            x = sch.scale_model_input(lat, t)
            io["sample"].tensor[:bs].copy_(x)
            io["sample"].tensor[bs:].copy_(x)
            io["t"].tensor.fill_(float(t))
            (p.prog if (i == 0 or p.prog_text_cached is None) else p.prog_text_cached).run(s)
            d = lib.CfgDdimDesc(eps=io["eps"].ptr, x=0, out=eps.data_ptr(), out2=0, nb=bs, chw=chw,
                                guidance=float(guidance_scale), do_step=0)
            lib.call(lib.OP_CFG_DDIM, d, s)
            lat = sch.step(eps, t, lat, generator=self.sched_generator).prev_sample
        if self.store is not None:
            eng.set_lora(False)
        return lat

    @torch.no_grad()
    def generate(self, ctx, noise, **kw) -> torch.Tensor:
        """-> uint8 images [bs][H][W][3] (needs a VaeDecoder)."""
        if self.decoder is None:
            raise RuntimeError("SliderSampler.generate needs a VaeDecoder")
        lat = self.sample_latents(ctx, noise, **kw)
        return VaeDecoder.to_uint8(self.decoder.decode(lat))
