"""TEST INFRASTRUCTURE ONLY - a SECOND, independently written restatement of the epsilon-network the reference calls
(`unet(...)` at trainscripts/textsliders/train_util.py:159-163 and 242-247; diffusers-0.20.2 UNet2DConditionModel, which is
neither vendored in the reference nor installable here).

Why it exists ("parity unpinned"): every epsilon in this repo is checked against oracle/unet_oracle.py, a
restatement written once, as an nn.Module tree.  An error in that restatement would be shared by the oracle and by the engine
built against it.  This file is derived from SURVEY.md Appendix A ALONE (hyper-parameter table, forward order, block formulas,
module names) - no nn.Module, no F.conv2d / F.group_norm / F.layer_norm / F.scaled_dot_product_attention / F.gelu: plain
tensor algebra in float64 over a flat {diffusers parameter name: tensor} dictionary - and tests/test_oracle_f64.py requires the two
restatements to agree to 1e-10 on the SD-1.x, SD-2.x and SDXL topologies.  Two independent derivations agreeing is the strongest
pin available offline; it is still not the reference's own arithmetic, so DESIGN.md keeps saying "parity unpinned".

Nothing under sliders_amd/ or bench.py's timed region may import this file.
"""
import math

import torch


def _conv(x, w, b, stride=1, pad=None):
    """Convolution as explicit patch extraction + matrix product (Appendix A: 3x3 p1 everywhere, 1x1 for shortcuts / SD-1.x proj)."""
    co, ci, kh, kw = w.shape
    pad = (kh // 2) if pad is None else pad
    B, C, H, W = x.shape
    assert C == ci
    xp = torch.zeros(B, C, H + 2 * pad, W + 2 * pad, dtype=x.dtype)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.zeros(B, co, Ho, Wo, dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, :, dy:dy + stride * (Ho - 1) + 1:stride, dx:dx + stride * (Wo - 1) + 1:stride]      # (B, ci, Ho, Wo)
            out += torch.einsum("bchw,oc->bohw", patch, w[:, :, dy, dx])
    return out + b.view(1, co, 1, 1)


def _linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def _silu(x):
    return x / (1.0 + torch.exp(-x))


def _gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _group_norm(x, w, b, groups, eps):
    B, C, H, W = x.shape
    g = x.reshape(B, groups, (C // groups) * H * W)
    mu = g.mean(-1, keepdim=True)
    var = ((g - mu) ** 2).mean(-1, keepdim=True)            # biased
    y = ((g - mu) / torch.sqrt(var + eps)).reshape(B, C, H, W)
    return y * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _sinusoid(t, dim):
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): computed in fp32 (Appendix A step 1), [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t[:, None].to(torch.float32) * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class FunctionalUNetF64:
    """cfg: any object with the Appendix-A fields (oracle.unet_oracle.UNetConfig works); sd: {name: tensor} in diffusers naming."""

    def __init__(self, cfg, sd):
        self.c = cfg
        self.p = {k: v.detach().to(torch.float64) for k, v in sd.items()}

    # ---- blocks -------------------------------------------------------------------------------------------------------
    def _gn(self, x, name, eps):
        return _group_norm(x, self.p[name + ".weight"], self.p[name + ".bias"], self.c.norm_num_groups, eps)

    def _lin(self, x, name, bias=True):
        return _linear(x, self.p[name + ".weight"], self.p[name + ".bias"] if bias else None)

    def _cv(self, x, name, stride=1):
        return _conv(x, self.p[name + ".weight"], self.p[name + ".bias"], stride)

    def _resnet(self, x, emb, name):
        h = self._cv(_silu(self._gn(x, name + ".norm1", self.c.norm_eps)), name + ".conv1")
        h = h + self._lin(_silu(emb), name + ".time_emb_proj")[:, :, None, None]
        h = self._cv(_silu(self._gn(h, name + ".norm2", self.c.norm_eps)), name + ".conv2")
        sc = self._cv(x, name + ".conv_shortcut") if (name + ".conv_shortcut.weight") in self.p else x
        return sc + h

    def _attn(self, x, ctx, name, heads):
        B, T, C = x.shape
        src = x if ctx is None else ctx
        q, k, v = self._lin(x, name + ".to_q", False), self._lin(src, name + ".to_k", False), self._lin(src, name + ".to_v", False)
        d = C // heads
        split = lambda z: z.reshape(B, z.shape[1], heads, d).permute(0, 2, 1, 3)          # (B, h, T, d)
        q, k, v = split(q), split(k), split(v)
        s = torch.einsum("bhqd,bhkd->bhqk", q, k) * (d ** -0.5)
        s = s - s.max(-1, keepdim=True).values
        pr = torch.exp(s)
        pr = pr / pr.sum(-1, keepdim=True)
        o = torch.einsum("bhqk,bhkd->bhqd", pr, v).permute(0, 2, 1, 3).reshape(B, T, C)
        return self._lin(o, name + ".to_out.0")

    def _tblock(self, x, ctx, name, heads):
        ln = lambda z, n: _layer_norm(z, self.p[f"{name}.{n}.weight"], self.p[f"{name}.{n}.bias"])
        x = x + self._attn(ln(x, "norm1"), None, name + ".attn1", heads)
        x = x + self._attn(ln(x, "norm2"), ctx, name + ".attn2", heads)
        pj = self._lin(ln(x, "norm3"), name + ".ff.net.0.proj")
        a, g = pj[..., : pj.shape[-1] // 2], pj[..., pj.shape[-1] // 2:]
        return x + self._lin(a * _gelu_erf(g), name + ".ff.net.2")

    def _transformer(self, x, ctx, name, layers, heads):
        B, C, H, W = x.shape
        res = x
        h = self._gn(x, name + ".norm", 1e-6)
        if self.c.use_linear_projection:
            h = self._lin(h.permute(0, 2, 3, 1).reshape(B, H * W, C), name + ".proj_in")
        else:
            h = self._cv(h, name + ".proj_in").permute(0, 2, 3, 1).reshape(B, H * W, C)
        for k in range(layers):
            h = self._tblock(h, ctx, f"{name}.transformer_blocks.{k}", heads)
        if self.c.use_linear_projection:
            h = self._lin(h, name + ".proj_out").reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self._cv(h.reshape(B, H, W, C).permute(0, 3, 1, 2), name + ".proj_out")
        return h + res

    # ---- the network --------------------------------------------------------------------------------------------------
    def __call__(self, sample, t, ctx, added=None):
        c = self.c
        f64 = torch.float64
        x, ctx = sample.to(f64), ctx.to(f64)
        B = x.shape[0]
        tt = torch.as_tensor(t).reshape(-1).expand(B)
        emb = self._lin(_silu(self._lin(_sinusoid(tt, c.block_out_channels[0]).to(f64), "time_embedding.linear_1")),
                        "time_embedding.linear_2")
        if c.addition_embed_type == "text_time":
            ids = added["time_ids"].reshape(-1)
            te = _sinusoid(ids, c.addition_time_embed_dim).reshape(B, -1)
            aug_in = torch.cat([added["text_embeds"].to(f64), te.to(f64)], dim=-1)
            emb = emb + self._lin(_silu(self._lin(aug_in, "add_embedding.linear_1")), "add_embedding.linear_2")
        boc, L = c.block_out_channels, c.layers_per_block
        h = self._cv(x, "conv_in")
        skips = [h]
        for i, typ in enumerate(c.down_block_types):
            for j in range(L):
                h = self._resnet(h, emb, f"down_blocks.{i}.resnets.{j}")
                if typ == "CrossAttnDownBlock2D":
                    h = self._transformer(h, ctx, f"down_blocks.{i}.attentions.{j}", c.transformer_layers_per_block[i],
                                          c.attention_head_dim[i])
                skips.append(h)
            if i != len(boc) - 1:
                h = self._cv(h, f"down_blocks.{i}.downsamplers.0.conv", stride=2)
                skips.append(h)
        h = self._resnet(h, emb, "mid_block.resnets.0")
        h = self._transformer(h, ctx, "mid_block.attentions.0", c.transformer_layers_per_block[-1], c.attention_head_dim[-1])
        h = self._resnet(h, emb, "mid_block.resnets.1")
        rl, rh = tuple(reversed(c.transformer_layers_per_block)), tuple(reversed(c.attention_head_dim))
        for i, typ in enumerate(c.up_block_types):
            for j in range(L + 1):
                h = self._resnet(torch.cat([h, skips.pop()], dim=1), emb, f"up_blocks.{i}.resnets.{j}")
                if typ == "CrossAttnUpBlock2D":
                    h = self._transformer(h, ctx, f"up_blocks.{i}.attentions.{j}", rl[i], rh[i])
            if i != len(boc) - 1:
                h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)           # nearest x 2
                h = self._cv(h, f"up_blocks.{i}.upsamplers.0.conv")
        assert not skips
        return self._cv(_silu(self._gn(h, "conv_norm_out", c.norm_eps)), "conv_out")
