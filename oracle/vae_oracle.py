"""CPU oracle: AutoencoderKL of Stable Diffusion (SD-1.x and SDXL share the architecture) as diffusers-0.20.2 builds
it, restated from the published architecture, plus `get_noisy_image` of the reference's image sliders.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
sliders_amd/ may import it.

What it follows
  * trainscripts/imagesliders/train_util.py:200-235 `get_noisy_image`: VaeImageProcessor.preprocess (PIL -> NCHW float in
    [-1, 1]) -> `vae.encode(image).latent_dist.sample()` -> `* vae.config.scaling_factor` -> `scheduler.add_noise(latents,
    noise, scheduler.timesteps[k])`; the VAE runs in fp32 (trainscripts/imagesliders/train_lora-scale-xl.py:96 moves it to
    the device without a dtype).
  * diffusers-0.20.2 `AutoencoderKL` / `Encoder` / `Decoder` / `DownEncoderBlock2D` / `UNetMidBlock2D` / `Attention` with
    the SD VAE config: block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4, norm_num_groups 32,
    act silu, GroupNorm eps 1e-6, ONE attention head of 512 channels in the mid block, Downsample2D(padding=0) = zero pad
    (right, bottom) by one then 3x3 stride-2 convolution, quant_conv 1x1, DiagonalGaussianDistribution with
    logvar clamped to [-30, 20].  scaling_factor 0.18215 (SD-1.x) / 0.13025 (SDXL).
PARITY UNPINNED for the arithmetic (diffusers is not vendored and not installed); pinned facts: the published parameter
count of the SD VAE (83,653,863) and the diffusers state-dict key names (tests/test_oracle.py).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

VAE_SCALING = {"sd1": 0.18215, "sdxl": 0.13025}


class VaeResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class VaeDownsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class VaeUpsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class VaeAttention(nn.Module):
    """diffusers Attention as used by UNetMidBlock2D of the VAE: GroupNorm, one head, biased projections, residual."""

    def __init__(self, c: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])
        self.scale = c ** -0.5

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        t = self.group_norm(x.view(b, c, h * w)).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.baddbmm(torch.empty(b, h * w, h * w, dtype=q.dtype, device=q.device), q, k.transpose(1, 2),
                                        beta=0, alpha=self.scale).float(), dim=-1).to(q.dtype)
        o = self.to_out[0](torch.bmm(p, v))
        return o.transpose(1, 2).reshape(b, c, h, w) + res


class _Mid(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(c, c), VaeResnetBlock2D(c, c)])
        self.attentions = nn.ModuleList([VaeAttention(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(cin, cout), VaeResnetBlock2D(cout, cout)])
        self.downsamplers = nn.ModuleList([VaeDownsample2D(cout)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock2D(cin if i == 0 else cout, cout) for i in range(3)])
        self.upsamplers = nn.ModuleList([VaeUpsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, boc=(128, 256, 512, 512), in_ch: int = 3, latent: int = 4):
        super().__init__()
        self.conv_in = nn.Conv2d(in_ch, boc[0], 3, padding=1)
        blocks, c = [], boc[0]
        for i, co in enumerate(boc):
            blocks.append(_DownBlock(c, co, i != len(boc) - 1))
            c = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, boc=(128, 256, 512, 512), out_ch: int = 3, latent: int = 4):
        super().__init__()
        rb = list(reversed(boc))
        self.conv_in = nn.Conv2d(latent, rb[0], 3, padding=1)
        self.mid_block = _Mid(rb[0])
        blocks, c = [], rb[0]
        for i, co in enumerate(rb):
            blocks.append(_UpBlock(c, co, i != len(rb) - 1))
            c = co
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, out_ch, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, moments: torch.Tensor):
        self.mean, self.logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * noise.to(self.mean)

    def mode(self):
        return self.mean


class _EncOut:
    def __init__(self, dist):
        self.latent_dist = dist


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class AutoencoderKL(nn.Module):
    def __init__(self, boc=(128, 256, 512, 512), scaling_factor: float = 0.18215, with_decoder: bool = True):
        super().__init__()
        self.encoder = Encoder(boc)
        self.quant_conv = nn.Conv2d(8, 8, 1)
        if with_decoder:
            self.decoder = Decoder(boc)
            self.post_quant_conv = nn.Conv2d(4, 4, 1)
        self.config = _Cfg(block_out_channels=tuple(boc), scaling_factor=scaling_factor, latent_channels=4)

    def encode(self, x):
        return _EncOut(DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def build_vae(kind: str = "sdxl", seed: int = 0, boc=(128, 256, 512, 512), with_decoder: bool = False) -> AutoencoderKL:
    """Seeded random-init VAE (no checkpoints exist offline)."""
    st = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        vae = AutoencoderKL(boc, VAE_SCALING[kind], with_decoder=with_decoder)
    finally:
        torch.random.set_rng_state(st)
    vae.requires_grad_(False)
    return vae.eval()


def preprocess_image(img) -> torch.Tensor:
    """VaeImageProcessor.preprocess for one PIL image / HxWx3 uint8 array: [0,255] -> float32 NCHW in [-1, 1]
    (sizes are multiples of 8 in the reference's loop: it resizes to 512 or 256 first)."""
    import numpy as np
    a = np.asarray(img, dtype=np.float32) / 255.0
    if a.ndim == 2:
        a = a[..., None].repeat(3, -1)
    t = torch.from_numpy(a[..., :3].copy()).permute(2, 0, 1)[None]
    return 2.0 * t - 1.0


def add_noise(alphas_cumprod: torch.Tensor, x0: torch.Tensor, noise: torch.Tensor, t: int) -> torch.Tensor:
    """DDIMScheduler.add_noise: sqrt(a_t) x0 + sqrt(1 - a_t) noise."""
    a = alphas_cumprod[t].to(x0.dtype)
    return a.sqrt() * x0 + (1 - a).sqrt() * noise


def get_noisy_image(img, vae: AutoencoderKL, alphas_cumprod: torch.Tensor, timestep: int, post_noise: torch.Tensor,
                    noise: torch.Tensor):
    """trainscripts/imagesliders/train_util.py:200-235 with the two random draws passed in explicitly (the reference
    draws them from the global / the seeded generator): posterior sample noise and the diffusion noise."""
    image = preprocess_image(img) if not torch.is_tensor(img) else img
    lat = vae.encode(image).latent_dist.sample(noise=post_noise)
    lat = vae.config.scaling_factor * lat
    return add_noise(alphas_cumprod, lat, noise, timestep), noise
