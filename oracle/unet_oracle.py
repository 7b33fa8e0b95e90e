"""CPU oracle: restatement of the UNet2DConditionModel the reference calls.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sliders_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, as the checker / CPU baseline.

What it restates
----------------
The reference (rohitgandikota/sliders) never defines the UNet: it calls
``unet(latent_model_input, timestep, encoder_hidden_states=...,
added_cond_kwargs=...).sample`` (trainscripts/textsliders/train_util.py:159-163,
242-247) on a ``diffusers==0.20.2`` ``UNet2DConditionModel``
(requirements.txt:3; loaded at trainscripts/textsliders/model_util.py:67-72,
169-174).  diffusers is a third-party dependency that is NOT vendored under
/root/reference and NOT installed in this image, so this file restates the
published diffusers-0.20.2 module graph for the two model families the
reference trains (SD-1.x and SDXL-base), following SURVEY.md Appendix A.

PARITY UNPINNED (UNet arithmetic): the reference ships no tests, golden vectors
or fixtures for this path and the real diffusers package cannot be imported
here, so the op order / hyper-parameters below are pinned only by (a) the
reference's own LoRA target census and state-dict key layout, which are
produced by running the reference's unmodified ``LoRANetwork``
(trainscripts/textsliders/lora.py:115-258) over THIS module tree (class names
and module paths are the diffusers ones on purpose) and (b) the known
parameter counts of the two architectures (859.5 M / 2567 M).

All modules are plain ``torch.nn`` and run on CPU in fp32 (the "truth" arm) or
bf16 (the "reference precision" arm).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# configs (diffusers `unet/config.json` of the checkpoints named in
# trainscripts/textsliders/data/config.yaml:3 and config-xl.yaml:3)
# --------------------------------------------------------------------------
@dataclass
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    # diffusers quirk: the config field `attention_head_dim` is used as the
    # NUMBER of heads.
    attention_head_dim: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: Optional[int] = None
    projection_class_embeddings_input_dim: Optional[int] = None
    time_embed_dim: int = 0  # filled in __post_init__ (4 * block_out_channels[0])

    def __post_init__(self):
        if not self.time_embed_dim:
            self.time_embed_dim = 4 * self.block_out_channels[0]


def sd1_config() -> UNetConfig:
    """CompVis/stable-diffusion-v1-4 and runwayml/stable-diffusion-v1-5."""
    return UNetConfig()


def sdxl_config() -> UNetConfig:
    """stabilityai/stable-diffusion-xl-base-1.0."""
    return UNetConfig(
        sample_size=128,
        block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 10),
        attention_head_dim=(5, 10, 20),
        cross_attention_dim=2048,
        use_linear_projection=True,
        addition_embed_type="text_time",
        addition_time_embed_dim=256,
        projection_class_embeddings_input_dim=2816,
    )


def sd2_config() -> UNetConfig:
    """stabilityai/stable-diffusion-2-1(-base) (the reference's pretrained_model.v2, model_util.py:31-50): 5/10/20/20 heads
    of width 64, Linear proj_in/proj_out, 1024-d context.  Published size: 865,910,724 parameters."""
    return UNetConfig(attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True)


def tiny_sd2_config() -> UNetConfig:
    """Same topology as SD-2.x, narrow."""
    return UNetConfig(sample_size=16, block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4),
                      cross_attention_dim=128, use_linear_projection=True)


def tiny_sd1_config() -> UNetConfig:
    """Same topology as SD-1.x, 1/5 width (channels stay multiples of 64) - for fast tests."""
    return UNetConfig(
        sample_size=16,
        block_out_channels=(64, 128, 320, 320),
        attention_head_dim=(8, 8, 8, 8),   # 8 heads like SD-1.x -> head dims 8 / 16 / 40 / 40
        cross_attention_dim=128,
    )


def tiny_sdxl_config() -> UNetConfig:
    """Same topology as SDXL (linear proj, text_time embedding, head_dim 64), narrow."""
    return UNetConfig(
        sample_size=16,
        block_out_channels=(64, 128, 256),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 1, 2),
        attention_head_dim=(1, 2, 4),
        cross_attention_dim=128,
        use_linear_projection=True,
        addition_embed_type="text_time",
        addition_time_embed_dim=32,
        projection_class_embeddings_input_dim=64 + 6 * 32,
    )


CONFIGS = {
    "sd1": sd1_config,
    "sd2": sd2_config,
    "tiny_sd2": tiny_sd2_config,
    "sdxl": sdxl_config,
    "tiny_sd1": tiny_sd1_config,
    "tiny_sdxl": tiny_sdxl_config,
}


# --------------------------------------------------------------------------
# leaf modules (class NAMES matter: the reference's LoRA target filter is by
# `module.__class__.__name__`, trainscripts/textsliders/lora.py:194-196)
# --------------------------------------------------------------------------
class LoRACompatibleConv(nn.Conv2d):
    """diffusers-0.20.x wraps resnet / sampler convs in this subclass."""


class LoRACompatibleLinear(nn.Linear):
    """diffusers-0.20.x wraps time_emb_proj / FF / proj_in/out linears in this subclass."""


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int,
                           flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0.0,
                           max_period: int = 10000) -> torch.Tensor:
    """Sinusoidal embedding, computed in fp32."""
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0.0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos,
                                      self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int = 1):
        super().__init__()
        self.conv = LoRACompatibleConv(channels, channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = LoRACompatibleConv(channels, channels, 3, padding=1)

    def forward(self, hidden_states):
        dtype = hidden_states.dtype
        if dtype == torch.bfloat16:  # diffusers upcasts: nearest is not implemented for bf16 there
            hidden_states = hidden_states.to(torch.float32)
        hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(dtype)
        return self.conv(hidden_states)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int,
                 groups: int = 32, eps: float = 1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = LoRACompatibleConv(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = LoRACompatibleLinear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = LoRACompatibleConv(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = LoRACompatibleConv(in_channels, out_channels, 1, stride=1, padding=0)

    def forward(self, input_tensor, temb):
        hidden_states = self.norm1(input_tensor)
        hidden_states = self.nonlinearity(hidden_states)
        hidden_states = self.conv1(hidden_states)
        temb = self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        hidden_states = hidden_states + temb
        hidden_states = self.norm2(hidden_states)
        hidden_states = self.nonlinearity(hidden_states)
        hidden_states = self.dropout(hidden_states)
        hidden_states = self.conv2(hidden_states)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + hidden_states) / 1.0


class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner_dim = dim_head * heads
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(kv_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(kv_dim, inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b, t, _ = hidden_states.shape
        q = self.to_q(hidden_states)
        k = self.to_k(ctx)
        v = self.to_v(ctx)
        h = self.heads
        q = q.view(b, t, h, -1).transpose(1, 2)
        k = k.view(b, ctx.shape[1], h, -1).transpose(1, 2)
        v = v.view(b, ctx.shape[1], h, -1).transpose(1, 2)
        # softmax(q k^T * scale) v ; the reference enables xformers (train_lora.py:68) which
        # computes the same function with fp32 softmax statistics.
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, t, -1)
        o = self.to_out[0](o)
        o = self.to_out[1](o)
        return o


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = LoRACompatibleLinear(dim_in, dim_out * 2)

    def forward(self, hidden_states):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        inner_dim = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner_dim), nn.Dropout(0.0),
                                  LoRACompatibleLinear(inner_dim, dim)])

    def forward(self, hidden_states):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class Transformer2DModel(nn.Module):
    def __init__(self, heads: int, dim_head: int, in_channels: int, num_layers: int,
                 cross_attention_dim: int, use_linear_projection: bool, groups: int = 32):
        super().__init__()
        inner_dim = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = LoRACompatibleLinear(in_channels, inner_dim)
        else:
            self.proj_in = LoRACompatibleConv(in_channels, inner_dim, 1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, heads, dim_head, cross_attention_dim)
            for _ in range(num_layers)])
        if use_linear_projection:
            self.proj_out = LoRACompatibleLinear(inner_dim, in_channels)
        else:
            self.proj_out = LoRACompatibleConv(inner_dim, in_channels, 1, stride=1, padding=0)

    def forward(self, hidden_states, encoder_hidden_states):
        b, c, h, w = hidden_states.shape
        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        if not self.use_linear_projection:
            hidden_states = self.proj_in(hidden_states)
            inner = hidden_states.shape[1]
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        else:
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, c)
            hidden_states = self.proj_in(hidden_states)
            inner = hidden_states.shape[-1]
        for block in self.transformer_blocks:
            hidden_states = block(hidden_states, encoder_hidden_states)
        if not self.use_linear_projection:
            hidden_states = hidden_states.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
            hidden_states = self.proj_out(hidden_states)
        else:
            hidden_states = self.proj_out(hidden_states)
            hidden_states = hidden_states.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return hidden_states + residual


# --------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------
class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, groups, eps)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, encoder_hidden_states=None):
        outs = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, tlayers, heads,
                 cross_attention_dim, use_linear_projection, add_downsample, groups, eps):
        super().__init__()
        resnets = [ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb_channels, groups, eps)
                   for i in range(num_layers)]
        # diffusers registers `attentions` before `resnets` in the cross-attention blocks; this fixes
        # the named_modules() order the reference's LoRANetwork.create_modules walks (lora.py:175)
        self.attentions = nn.ModuleList([
            Transformer2DModel(heads, out_channels // heads, out_channels, tlayers, cross_attention_dim,
                               use_linear_projection, groups)
            for _ in range(num_layers)])
        self.resnets = nn.ModuleList(resnets)
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, encoder_hidden_states=None):
        outs = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb)
            hidden_states = attn(hidden_states, encoder_hidden_states)
            outs += (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            outs += (hidden_states,)
        return hidden_states, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, tlayers, heads, cross_attention_dim,
                 use_linear_projection, groups, eps):
        super().__init__()
        resnets = [ResnetBlock2D(in_channels, in_channels, temb_channels, groups, eps),
                   ResnetBlock2D(in_channels, in_channels, temb_channels, groups, eps)]
        self.attentions = nn.ModuleList([
            Transformer2DModel(heads, in_channels // heads, in_channels, tlayers, cross_attention_dim,
                               use_linear_projection, groups)])
        self.resnets = nn.ModuleList(resnets)

    def forward(self, hidden_states, temb, encoder_hidden_states=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states)
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 add_upsample, groups, eps):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(resnet_in_channels + res_skip_channels, out_channels,
                                         temb_channels, groups, eps))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states=None):
        for resnet in self.resnets:
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 tlayers, heads, cross_attention_dim, use_linear_projection, add_upsample, groups, eps):
        super().__init__()
        resnets, attentions = [], []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(resnet_in_channels + res_skip_channels, out_channels,
                                         temb_channels, groups, eps))
            attentions.append(Transformer2DModel(heads, out_channels // heads, out_channels, tlayers,
                                                 cross_attention_dim, use_linear_projection, groups))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states=None):
        for resnet, attn in zip(self.resnets, self.attentions):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
            hidden_states = attn(hidden_states, encoder_hidden_states)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


# --------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------
class UNetOutput:
    """Stand-in for diffusers' UNet2DConditionOutput: `.sample` and tuple indexing."""

    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class _Config(dict):
    __getattr__ = dict.__getitem__


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        self.config = _Config(in_channels=cfg.in_channels, sample_size=cfg.sample_size,
                              addition_time_embed_dim=cfg.addition_time_embed_dim)
        self.in_channels = cfg.in_channels
        boc = cfg.block_out_channels
        ted = cfg.time_embed_dim
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], True, 0.0)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        if cfg.addition_embed_type == "text_time":
            self.add_time_proj = Timesteps(cfg.addition_time_embed_dim, True, 0.0)
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted)

        # diffusers creates both ModuleLists up front, so `up_blocks` precedes `mid_block` in named_modules()
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        output_channel = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            input_channel, output_channel = output_channel, boc[i]
            is_final = i == len(boc) - 1
            if t == "DownBlock2D":
                blk = DownBlock2D(input_channel, output_channel, ted, cfg.layers_per_block,
                                  not is_final, g, eps)
            else:
                blk = CrossAttnDownBlock2D(input_channel, output_channel, ted, cfg.layers_per_block,
                                           cfg.transformer_layers_per_block[i], cfg.attention_head_dim[i],
                                           cfg.cross_attention_dim, cfg.use_linear_projection,
                                           not is_final, g, eps)
            self.down_blocks.append(blk)

        self.mid_block = UNetMidBlock2DCrossAttn(
            boc[-1], ted, cfg.transformer_layers_per_block[-1], cfg.attention_head_dim[-1],
            cfg.cross_attention_dim, cfg.use_linear_projection, g, eps)

        rboc = tuple(reversed(boc))
        rheads = tuple(reversed(cfg.attention_head_dim))
        rtl = tuple(reversed(cfg.transformer_layers_per_block))
        output_channel = rboc[0]
        for i, t in enumerate(cfg.up_block_types):
            is_final = i == len(boc) - 1
            prev_output_channel, output_channel = output_channel, rboc[i]
            input_channel = rboc[min(i + 1, len(boc) - 1)]
            if t == "UpBlock2D":
                blk = UpBlock2D(input_channel, prev_output_channel, output_channel, ted,
                                cfg.layers_per_block + 1, not is_final, g, eps)
            else:
                blk = CrossAttnUpBlock2D(input_channel, prev_output_channel, output_channel, ted,
                                         cfg.layers_per_block + 1, rtl[i], rheads[i],
                                         cfg.cross_attention_dim, cfg.use_linear_projection,
                                         not is_final, g, eps)
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def enable_xformers_memory_efficient_attention(self):  # train_lora.py:68 calls it
        pass

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None,
                return_dict: bool = True):
        cfg = self.cfg
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None]
        timesteps = timesteps.expand(sample.shape[0])
        t_emb = self.time_proj(timesteps).to(dtype=sample.dtype)
        emb = self.time_embedding(t_emb)
        if cfg.addition_embed_type == "text_time":
            text_embeds = added_cond_kwargs["text_embeds"]
            time_ids = added_cond_kwargs["time_ids"]
            time_embeds = self.add_time_proj(time_ids.flatten())
            time_embeds = time_embeds.reshape((text_embeds.shape[0], -1))
            add_embeds = torch.cat([text_embeds, time_embeds.to(text_embeds.dtype)], dim=-1).to(emb.dtype)
            emb = emb + self.add_embedding(add_embeds)

        sample = self.conv_in(sample)
        down_res = (sample,)
        for blk in self.down_blocks:
            sample, res = blk(sample, emb, encoder_hidden_states)
            down_res += res
        sample = self.mid_block(sample, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res = down_res[-n:]
            down_res = down_res[:-n]
            sample = blk(sample, res, emb, encoder_hidden_states)
        sample = self.conv_norm_out(sample)
        sample = self.conv_act(sample)
        sample = self.conv_out(sample)
        if not return_dict:
            return (sample,)
        return UNetOutput(sample)


def build_unet(name_or_cfg, seed: int = 0, dtype=torch.float32, device="cpu") -> UNet2DConditionModel:
    """Seeded random-init UNet (no checkpoints exist offline)."""
    cfg = CONFIGS[name_or_cfg]() if isinstance(name_or_cfg, str) else name_or_cfg
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        if device == "meta":
            with torch.device("meta"):
                net = UNet2DConditionModel(cfg)
        else:
            net = UNet2DConditionModel(cfg)
    finally:
        torch.random.set_rng_state(gen_state)
    net = net.to(dtype=dtype) if device != "meta" else net
    net.requires_grad_(False)
    net.eval()
    return net
