"""Fused image-slider training iteration (trainscripts/imagesliders/train_lora-scale-xl.py:178-396,
train_lora-scale.py:175-351) on one MI355X, data-parallel across N of them like the text sliders.

One iteration of the reference:
    k ~ U{1..49}; |scale| and an image pair (same file name in the -scale and +scale folders), resized to 512 (XL) / 256
    low, high = get_noisy_image(img, vae, ...)            VAE encode (fp32) -> * scaling_factor -> add_noise at timesteps_50[k],
                                                          the SAME seed for both images (train_util.py:200-235)
    t = timesteps_1000[int(k*1000/50)]
    [two no-grad predictions whose results are never used - skipped here, SURVEY.md D.12]
    set_lora_slider(+scale); with network: eps = predict_noise(unconditional|positive, high, guidance 1)
    loss_high = MSE(eps.float(), noise.float()); loss_high.backward()
    set_lora_slider(-scale); with network: eps = predict_noise(unconditional|neutral, low, guidance 1)
    loss_low = MSE(eps.float(), noise.float()); loss_low.backward()         # gradients ACCUMULATE
    optimizer.step()
Here: the VAE encoder is a command buffer of fp32 HIP kernels (sliders_amd/vae.py), each polarity is one training
forward + the backward program into the same flat fp32 gradient buffer, then (one all-reduce and) the fused AdamW.
The loss gradient 2/n (eps - noise), rounded to bf16 like autograd's cast back through `.to(float32)`, is produced by
slh_guidance_loss with positive = unconditional (its guidance term is then exactly zero).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import lib
from .trainer import PairEmbeds, SliderTrainer, _stream
from .vae import VaeEncoder


class ImageSliderTrainer(SliderTrainer):
    def __init__(self, engine, store, vae: VaeEncoder, H: int, W: int, **kw):
        super().__init__(engine, store, H, W, **kw)
        self.vae = vae
        self.loss_high = torch.zeros(1, dtype=torch.float32, device=engine.device)
        self.loss_low = torch.zeros(1, dtype=torch.float32, device=engine.device)

    def _polarity(self, pair_ctx, pair_pooled, image, post_noise, noise, noise_bf16, coeff, t_cur, multiplier, loss_out):
        eng, bs = self.eng, self.bs
        noisy, _, _ = self.vae.get_noisy_image(image, post_noise, noise, coeff[0], coeff[1])
        eng.set_lora(True, multiplier)                       # network.set_lora_slider(scale); with network: ...
        p_tr = eng.plan(2 * bs, self.H, self.W, "train")
        self._predict(p_tr, noisy, pair_ctx, pair_pooled, t_cur, self.e_tgt)
        loss_out.zero_()
        bw = p_tr.backward
        nb = noise_bf16.data_ptr()
        d = lib.LossDesc(target=self.e_tgt.data_ptr(), positive=nb, neutral=nb, uncond=nb, loss=loss_out.data_ptr(),
                         dtarget=0, dtarget_pix=bw.deps_pix.ptr, n=bs * self.chw, guidance=1.0, erase=0,
                         hw=self.H * self.W, nch=eng.cfg.out_channels)
        lib.call(lib.OP_LOSS, d, _stream())
        bw.prog.run(_stream())                               # accumulates into store.grads

    def iteration(self, pair: PairEmbeds, k: int, img_low: torch.Tensor, img_high: torch.Tensor, scale: float,
                  post_noise: torch.Tensor, noise: torch.Tensor, lr: Optional[float] = None,
                  time_ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """img_*: [bs][H*8][W*8][3] float32 in [-1,1] (VaeEncoder.preprocess); post_noise / noise: (bs,4,H,W) float32,
        shared by the two images like the reference's re-seeded generator.  Returns (loss_high, loss_low)."""
        bs = noise.shape[0]
        self._use(bs, noise.shape[2], noise.shape[3])
        if time_ids is not None:
            self.time_ids = time_ids.to(device=self.eng.device, dtype=torch.float32).reshape(2 * bs, 6)
        if lr is not None:
            self.lr = float(lr)
        st = self.store
        coeff = self.sched.add_noise_coefficients(self.t50[k])               # scheduler.timesteps[timesteps_to], 50 steps
        t_cur = self.t1000[int(k * 1000 / self.nsteps)]
        noise_bf16 = noise.to(self.eng.device, torch.bfloat16).contiguous()  # high_noise.to(device, dtype=weight_dtype)
        st.grads.zero_()
        self._polarity(pair.ctx_positive, pair.pooled_positive, img_high, post_noise, noise, noise_bf16, coeff, t_cur,
                       float(scale), self.loss_high)
        self._polarity(pair.ctx_neutral, pair.pooled_neutral, img_low, post_noise, noise, noise_bf16, coeff, t_cur,
                       -float(scale), self.loss_low)
        self.eng.set_lora(False)
        self.reduce_and_step()
        return self.loss_high, self.loss_low
