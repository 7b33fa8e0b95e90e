"""AutoencoderKL encoder on the MI355X for the image sliders (trainscripts/imagesliders/train_util.py:200-235
`get_noisy_image`): VaeImageProcessor.preprocess -> vae.encode(...).latent_dist.sample() -> * scaling_factor ->
scheduler.add_noise, all on the GPU in fp32 (the reference keeps the VAE in fp32,
trainscripts/imagesliders/train_lora-scale-xl.py:96).

The encoder is a static command buffer over the fp32 kernels of csrc/vae.hip (slh_sgemm implicit-GEMM convolutions on
the exact-fp32 MFMA, slh_gn32_*, slh_softmax32, slh_vae_conv_in, slh_vae_moments); the posterior sample + add_noise is
one more launch (slh_vae_sample) whose coefficients change every iteration.  Activations are pixel-major fp32
[B*H*W][C]; weights are repacked once from the diffusers AutoencoderKL state dict (`encoder.*`, `quant_conv.*`).
There is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import lib
from .arena import Arena, Buf

VAE_SCALING = {"sd1": 0.18215, "sdxl": 0.13025}       # vae.config.scaling_factor of the SD-1.x / SDXL checkpoints


def random_vae_state_dict(boc=(128, 256, 512, 512), device="cpu", seed: int = 0, decoder: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded random-init VAE weights with the diffusers key names (no checkpoints exist offline): PyTorch default init
    bounds, norms 1/0.  Encoder + quant_conv, and with `decoder` also decoder + post_quant_conv."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def u(shape, fan_in):
        b = 1.0 / fan_in ** 0.5
        return (torch.rand(shape, generator=g, device=device) * 2 - 1) * b

    def conv(name, co, ci, k):
        sd[name + ".weight"] = u((co, ci, k, k), ci * k * k)
        sd[name + ".bias"] = u((co,), ci * k * k)

    def lin(name, co, ci):
        sd[name + ".weight"] = u((co, ci), ci)
        sd[name + ".bias"] = u((co,), ci)

    def norm(name, c):
        sd[name + ".weight"] = torch.ones(c, device=device)
        sd[name + ".bias"] = torch.zeros(c, device=device)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, 3)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1)

    conv("encoder.conv_in", boc[0], 3, 3)
    c = boc[0]
    for i, co in enumerate(boc):
        resnet(f"encoder.down_blocks.{i}.resnets.0", c, co)
        resnet(f"encoder.down_blocks.{i}.resnets.1", co, co)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
        c = co
    resnet("encoder.mid_block.resnets.0", c, c)
    resnet("encoder.mid_block.resnets.1", c, c)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", c)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{nm}", c, c)
    norm("encoder.conv_norm_out", c)
    conv("encoder.conv_out", 8, c, 3)
    conv("quant_conv", 8, 8, 1)
    if decoder:
        rb = list(reversed(boc))
        conv("decoder.conv_in", rb[0], 4, 3)
        resnet("decoder.mid_block.resnets.0", rb[0], rb[0])
        resnet("decoder.mid_block.resnets.1", rb[0], rb[0])
        a = "decoder.mid_block.attentions.0"
        norm(a + ".group_norm", rb[0])
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{a}.{nm}", rb[0], rb[0])
        c = rb[0]
        for i, co in enumerate(rb):
            for j in range(3):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", c if j == 0 else co, co)
            if i != len(rb) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
            c = co
        norm("decoder.conv_norm_out", c)
        conv("decoder.conv_out", 3, c, 3)
        conv("post_quant_conv", 4, 4, 1)
    return sd


class _Plan:
    def __init__(self, prog, io, arena):
        self.prog, self.io, self.arena = prog, io, arena


# The vae/diffusion_pytorch_model.safetensors of CompVis/stable-diffusion-v1-4 and runwayml/stable-diffusion-v1-5 (the
# reference's SD-1.x defaults) still carry the mid-block attention under its deprecated names; diffusers renames them at load
# time (modeling_utils._convert_deprecated_attention_blocks), and so must anything that reads the file directly.
_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
_ATTN_W = tuple(f"attentions.0.{n}.weight" for n in ("to_q", "to_k", "to_v", "to_out.0"))


def normalize_vae_key(k: str) -> str:
    parts = k.split(".")
    if len(parts) >= 2 and "attentions" in parts and parts[-2] in _DEPRECATED_ATTN:
        parts[-2] = _DEPRECATED_ATTN[parts[-2]]
        return ".".join(parts)
    return k


class _VaeNet:
    """Weights + the op emitters shared by the encoder and the decoder command buffers."""
    PREFIXES: Tuple[str, ...] = ()

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0", scaling_factor: float = VAE_SCALING["sdxl"],
                 exact_fp32: bool = True):
        """exact_fp32 (the default: the reference declares its VAE fp32, imagesliders/train_lora-scale-xl.py:96): every product on
        the exact-fp32 matrix instruction (slh_sgemm_desc.split_bf16 = 0; 4e-6 relative to the fp32 oracle).  exact_fp32 = False is
        an explicit opt-in to narrower arithmetic: operands split into two bf16 halves (16 mantissa bits, three bf16 MFMAs per
        product block, fp32 accumulation, ~2e-5 relative to the fp32 result) at less than half the time."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} needs a ROCm GPU (there is no CPU fallback)")
        lib.load()
        self.split_bf16 = 0 if exact_fp32 else 1
        self.scaling_factor = float(scaling_factor)
        self.w: Dict[str, torch.Tensor] = {}
        f = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        c3 = lambda t: f(t.permute(0, 2, 3, 1).reshape(t.shape[0], -1))        # [Cout][Cin][3][3] -> [Cout][ky][kx][Cin]
        for k, v in state_dict.items():
            if not k.startswith(self.PREFIXES):
                continue
            k = normalize_vae_key(k)
            if k.endswith(_ATTN_W) and v.ndim == 4:       # pre-0.18 checkpoints keep the attention projections as 1x1 convs
                v = v.reshape(v.shape[0], v.shape[1])
            if k.endswith(".weight") and v.ndim == 4 and v.shape[2] == 3:
                self.w[k] = c3(v)
            elif k.endswith(".weight") and v.ndim == 4:
                self.w[k] = f(v.reshape(v.shape[0], v.shape[1]))
            else:
                self.w[k] = f(v)
        self._plans: Dict[Tuple[int, int, int], _Plan] = {}

    def nbytes(self) -> int:
        return sum(t.numel() * 4 for t in self.w.values())

    # ---- op emitters (B, prog, arena, zarena are set by _build) ------------------------------------------------------
    def _begin(self, B, arena, zarena):
        self._B, self._prog, self._arena, self._zarena = B, lib.Program(), arena, zarena
        self._wp = (lambda k: self.w[k].data_ptr()) if not arena.virtual else (lambda k: 0x1000)

    def _act(self, rows, C, name):
        return self._arena.alloc((rows, C), torch.float32, name)

    def _gn(self, x: Buf, C, hw, name, silu):
        B = self._B
        stats = self._arena.alloc((B, 32, 2), torch.float32, name + ".stats")
        prow, ntick = lib.gn32_workspace(hw)
        part = self._arena.alloc((B, prow, 32, 2), torch.float32, name + ".partial")
        ticket = self._zarena.alloc((B, ntick), torch.float32, name + ".ticket")      # arrival counters of the fixed-order reduction
        y = self._act(B * hw, C, name)
        d = lib.Gn32Desc(x=x.ptr, gamma=self._wp(name + ".weight"), beta=self._wp(name + ".bias"), stats=stats.ptr, y=y.ptr,
                         ldx=C, ldy=C, C=C, batch=B, hw=hw, groups=32, eps=1e-6, act=1 if silu else 0,
                         partial=part.ptr, ticket=ticket.ptr)
        self._prog.add(lib.OP_GN32_STATS, d, name + ".stats")
        self._prog.add(lib.OP_GN32_APPLY, d, name + ".apply")
        return y

    def _conv3(self, x: Buf, ci, co, h, w, name, stride=1, residual: Optional[Buf] = None, upsample=False, w_ptr=None,
               b_ptr=None):
        B = self._B
        if upsample:
            ho, wo = 2 * h, 2 * w
        else:
            ho, wo = (h, w) if stride == 1 else ((h - 2) // 2 + 1, (w - 2) // 2 + 1)   # stride 2: pad (0,1,0,1) then k3 s2
        y = self._act(B * ho * wo, co, name)
        d = lib.SgemmDesc(x=x.ptr, w=w_ptr or self._wp(name + ".weight"), bias=b_ptr or self._wp(name + ".bias"),
                          residual=residual.ptr if residual else 0, c=y.ptr, ldx=ci, ldw=9 * ci, ldr=co, ldc=co,
                          M=B * ho * wo, N=co, K=9 * ci, mode=1, cin=ci, batch=B, hs=h, ws=w, ho=ho, wo=wo, stride=stride,
                          pad=1 if stride == 1 else 0, alpha=1.0, upsample=1 if upsample else 0, split_bf16=self.split_bf16)
        self._prog.add(lib.OP_SGEMM, d, name)
        return y, ho, wo

    def _dense(self, x_ptr, M, K, w_ptr, N, bias_ptr, name, residual: Optional[Buf] = None, out: Optional[Buf] = None,
               alpha=1.0, bias_per_row=0):
        y = out or self._act(M, N, name)
        d = lib.SgemmDesc(x=x_ptr, w=w_ptr, bias=bias_ptr, residual=residual.ptr if residual else 0, c=y.ptr, ldx=K, ldw=K,
                          ldr=N, ldc=N, M=M, N=N, K=K, mode=0, alpha=alpha, bias_per_row=bias_per_row,
                          split_bf16=self.split_bf16)
        self._prog.add(lib.OP_SGEMM, d, name)
        return y

    def _resnet(self, x: Buf, ci, co, h, w, name):
        wp = self._wp
        a1 = self._gn(x, ci, h * w, name + ".norm1", True)
        h1, _, _ = self._conv3(a1, ci, co, h, w, name + ".conv1")
        a2 = self._gn(h1, co, h * w, name + ".norm2", True)
        sc = x
        if ci != co:
            sc = self._dense(x.ptr, self._B * h * w, ci, wp(name + ".conv_shortcut.weight"), co, wp(name + ".conv_shortcut.bias"),
                             name + ".conv_shortcut")
        out, _, _ = self._conv3(a2, co, co, h, w, name + ".conv2", residual=sc)
        return out

    def _attention(self, x: Buf, c, T, a):
        """UNetMidBlock2D attention: GroupNorm, ONE head of c channels, biased projections, residual."""
        B, wp = self._B, self._wp
        t = self._gn(x, c, T, a + ".group_norm", False)
        q = self._dense(t.ptr, B * T, c, wp(a + ".to_q.weight"), c, wp(a + ".to_q.bias"), a + ".to_q")
        k = self._dense(t.ptr, B * T, c, wp(a + ".to_k.weight"), c, wp(a + ".to_k.bias"), a + ".to_k")
        o = self._act(B * T, c, a + ".pv")
        for b in range(B):
            off = 4 * b * T * c
            # V^T [c][T] = Wv . t_b^T (+ bias per row): the P.V product then reads it as an [N][K] operand
            vt = self._dense(wp(a + ".to_v.weight"), c, c, t.ptr + off, T, wp(a + ".to_v.bias"), f"{a}.to_v_T.{b}", bias_per_row=1)
            s = self._dense(q.ptr + off, T, c, k.ptr + off, T, 0, f"{a}.scores.{b}", alpha=float(c) ** -0.5)
            self._prog.add(lib.OP_SOFTMAX32, lib.Softmax32Desc(x=s.ptr, ld=T, rows=T, cols=T), f"{a}.softmax.{b}")
            ob = Buf(o.ptr + off, 4 * T * c, (T, c), torch.float32, None, "")
            self._dense(s.ptr, T, T, vt.ptr, c, 0, f"{a}.pv.{b}", out=ob)
        return self._dense(o.ptr, B * T, c, wp(a + ".to_out.0.weight"), c, wp(a + ".to_out.0.bias"), a + ".to_out.0", residual=x)

    def _finish(self, io) -> _Plan:
        head = lib.Program()
        if self._zarena.mark() > 0:
            pz, nz = self._zarena.region(0, self._zarena.mark())
            head.memset(pz, nz, 0, "zero_gn_stats")
        head.extend(self._prog)
        return _Plan(head, io, self._arena)

    def _plan_for(self, key, build):
        p = self._plans.get(key)
        if p is None:
            va, vz = Arena(1 << 50, None, "vae-virtual"), Arena(1 << 40, None, "vae-virtual-z")
            build(va, vz)
            arena = Arena(va.high_water + (1 << 20), self.device, "vae activations")
            zarena = Arena(vz.high_water + 4096, self.device, "vae GroupNorm statistics")
            p = self._plans[key] = build(arena, zarena)
            p.zarena = zarena
        return p


class VaeEncoder(_VaeNet):
    PREFIXES = ("encoder.", "quant_conv.")

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0", scaling_factor: float = VAE_SCALING["sdxl"],
                 exact_fp32: bool = True):
        super().__init__(state_dict, device, scaling_factor, exact_fp32)
        self.boc = []
        i = 0
        while f"encoder.down_blocks.{i}.resnets.0.conv1.weight" in state_dict:
            self.boc.append(int(state_dict[f"encoder.down_blocks.{i}.resnets.0.conv1.weight"].shape[0]))
            i += 1
        if not self.boc:
            raise KeyError("state dict has no encoder.down_blocks.*: expected the diffusers AutoencoderKL layout")

    # ---- planner ------------------------------------------------------------------------------------------------
    def _build(self, B: int, H: int, W: int, arena: Arena, zarena: Arena) -> _Plan:
        self._begin(B, arena, zarena)
        prog, wp, f32 = self._prog, self._wp, torch.float32
        img = arena.alloc((B, H * W, 3), f32, "in.image")
        boc = self.boc
        x = self._act(B * H * W, boc[0], "encoder.conv_in")
        prog.add(lib.OP_VAE_CONV_IN, lib.VaeConvDesc(x=img.ptr, w=wp("encoder.conv_in.weight"), bias=wp("encoder.conv_in.bias"),
                                                     y=x.ptr, batch=B, h=H, wd=W, cin=3, cout=boc[0]), "encoder.conv_in")
        c, h, w = boc[0], H, W
        for i, co in enumerate(boc):
            p = f"encoder.down_blocks.{i}"
            x = self._resnet(x, c, co, h, w, p + ".resnets.0")
            x = self._resnet(x, co, co, h, w, p + ".resnets.1")
            c = co
            if i != len(boc) - 1:
                x, h, w = self._conv3(x, c, c, h, w, p + ".downsamplers.0.conv", stride=2)
        x = self._resnet(x, c, c, h, w, "encoder.mid_block.resnets.0")
        x = self._attention(x, c, h * w, "encoder.mid_block.attentions.0")
        x = self._resnet(x, c, c, h, w, "encoder.mid_block.resnets.1")
        g = self._gn(x, c, h * w, "encoder.conv_norm_out", True)
        mom = arena.alloc((B * h * w, 8), f32, "moments")
        prog.add(lib.OP_VAE_MOMENTS, lib.VaeConvDesc(x=g.ptr, w=wp("encoder.conv_out.weight"), bias=wp("encoder.conv_out.bias"),
                                                     qw=wp("quant_conv.weight"), qb=wp("quant_conv.bias"), y=mom.ptr,
                                                     batch=B, h=h, wd=w, cin=c, cout=8), "conv_out+quant_conv")
        io = {"image": img, "moments": mom, "h": h, "w": w,
              "post_noise": arena.alloc((B, 4, h * w), f32, "in.post_noise"),
              "noise": arena.alloc((B, 4, h * w), f32, "in.noise"),
              "latent": arena.alloc((B, 4, h, w), f32, "out.latent"),
              "noisy": arena.alloc((B, 4, h, w), f32, "out.noisy"),
              "noisy_bf16": arena.alloc((B, 4, h, w), torch.bfloat16, "out.noisy_bf16")}
        return self._finish(io)

    def plan(self, B: int, H: int, W: int) -> _Plan:
        if H % 8 or W % 8:
            raise ValueError("image sides must be multiples of 8 (the reference resizes to 512 / 256 first)")
        return self._plan_for((B, H, W), lambda a, z: self._build(B, H, W, a, z))

    # ---- the operator ---------------------------------------------------------------------------------------------
    @staticmethod
    def preprocess(img) -> torch.Tensor:
        """VaeImageProcessor.preprocess of one image: PIL / HxWx3 uint8 array / uint8 tensor -> float32 [1][H][W][3] in
        [-1, 1] (channels last = the pixel-major layout the encoder reads)."""
        if not torch.is_tensor(img):
            import numpy as np
            img = torch.from_numpy(np.asarray(img).copy())
        if img.ndim == 2:
            img = img[..., None].expand(-1, -1, 3)
        return (img[..., :3].to(torch.float32) / 255.0 * 2.0 - 1.0)[None]

    def encode_moments(self, image: torch.Tensor) -> torch.Tensor:
        """image: [B][H][W][3] float32 in [-1,1] -> moments [B*h*w][8] (mean | logvar), h = H/8."""
        B, H, W, _ = image.shape
        p = self.plan(B, H, W)
        p.io["image"].tensor.copy_(image.reshape(B, H * W, 3))
        p.prog.run(torch.cuda.current_stream().cuda_stream)
        return p.io["moments"].tensor

    def get_noisy_image(self, image: torch.Tensor, post_noise: torch.Tensor, noise: torch.Tensor, sqrt_alpha: float,
                        sqrt_one_minus_alpha: float):
        """(noisy latents bf16 (B,4,h,w), noisy latents fp32, clean scaled latents fp32).  post_noise / noise: (B,4,h,w)
        float32: the draw inside latent_dist.sample() and the diffusion noise of scheduler.add_noise."""
        B, H, W, _ = image.shape
        p = self.plan(B, H, W)
        self.encode_moments(image)
        io = p.io
        io["post_noise"].tensor.copy_(post_noise.reshape(B, 4, -1))
        io["noise"].tensor.copy_(noise.reshape(B, 4, -1))
        d = lib.VaeSampleDesc(moments=io["moments"].ptr, post_noise=io["post_noise"].ptr, noise=io["noise"].ptr,
                              latent_f32=io["latent"].ptr, noisy_f32=io["noisy"].ptr, noisy_bf16=io["noisy_bf16"].ptr,
                              batch=B, hw=io["h"] * io["w"], scaling=self.scaling_factor, sqrt_alpha=float(sqrt_alpha),
                              sqrt_one_minus_alpha=float(sqrt_one_minus_alpha))
        lib.call(lib.OP_VAE_SAMPLE, d, torch.cuda.current_stream().cuda_stream)
        return io["noisy_bf16"].tensor, io["noisy"].tensor, io["latent"].tensor


class VaeDecoder(_VaeNet):
    """`vae.decode(latents / scaling_factor).sample` of the slider inference path (eval-scripts/generate_images_sd1.py:
    166-170, trainscripts/textsliders/generate_images_xl.py): post_quant_conv -> Decoder (conv_in, mid block with one
    attention head, 4 up blocks of 3 resnets with nearest-2x Upsample2D folded into the following convolution's addressing,
    GroupNorm + SiLU, conv_out) in fp32 -> image in [-1, 1]."""
    PREFIXES = ("decoder.", "post_quant_conv.")

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0", scaling_factor: float = VAE_SCALING["sdxl"],
                 exact_fp32: bool = True):
        super().__init__(state_dict, device, scaling_factor, exact_fp32)
        self.boc = []          # decoder order: widest first
        i = 0
        while f"decoder.up_blocks.{i}.resnets.0.conv1.weight" in state_dict:
            self.boc.append(int(state_dict[f"decoder.up_blocks.{i}.resnets.0.conv1.weight"].shape[0]))
            i += 1
        if not self.boc:
            raise KeyError("state dict has no decoder.up_blocks.*: expected the diffusers AutoencoderKL layout")
        # conv_out has 3 output channels; slh_sgemm wants N % 4 == 0: one zero row / zero bias is appended
        w, b = self.w["decoder.conv_out.weight"], self.w["decoder.conv_out.bias"]
        self.w["decoder.conv_out.weight4"] = torch.cat([w, w.new_zeros(1, w.shape[1])]).contiguous()
        self.w["decoder.conv_out.bias4"] = torch.cat([b, b.new_zeros(1)]).contiguous()

    def _build(self, B: int, h: int, w: int, arena: Arena, zarena: Arena, z_bf16: bool) -> _Plan:
        self._begin(B, arena, zarena)
        prog, wp, f32 = self._prog, self._wp, torch.float32
        z = arena.alloc((B, 4, h * w), torch.bfloat16 if z_bf16 else f32, "in.latents")
        zq = self._act(B * h * w, 4, "post_quant_conv")
        prog.add(lib.OP_VAE_POST_QUANT, lib.VaeConvDesc(x=z.ptr, qw=wp("post_quant_conv.weight"), qb=wp("post_quant_conv.bias"),
                                                        y=zq.ptr, batch=B, h=h, wd=w, cin=1 if z_bf16 else 0, cout=4,
                                                        inv_scaling=1.0 / self.scaling_factor), "post_quant_conv")
        c = self.boc[0]
        x = self._act(B * h * w, c, "decoder.conv_in")
        prog.add(lib.OP_VAE_CONV_IN, lib.VaeConvDesc(x=zq.ptr, w=wp("decoder.conv_in.weight"), bias=wp("decoder.conv_in.bias"),
                                                     y=x.ptr, batch=B, h=h, wd=w, cin=4, cout=c), "decoder.conv_in")
        x = self._resnet(x, c, c, h, w, "decoder.mid_block.resnets.0")
        x = self._attention(x, c, h * w, "decoder.mid_block.attentions.0")
        x = self._resnet(x, c, c, h, w, "decoder.mid_block.resnets.1")
        for i, co in enumerate(self.boc):
            p = f"decoder.up_blocks.{i}"
            for j in range(3):
                x = self._resnet(x, c if j == 0 else co, co, h, w, f"{p}.resnets.{j}")
            c = co
            if i != len(self.boc) - 1:
                x, h, w = self._conv3(x, c, c, h, w, p + ".upsamplers.0.conv", upsample=True)
        g = self._gn(x, c, h * w, "decoder.conv_norm_out", True)
        img, _, _ = self._conv3(g, c, 4, h, w, "decoder.conv_out", w_ptr=wp("decoder.conv_out.weight4"),
                                b_ptr=wp("decoder.conv_out.bias4"))
        return self._finish({"latents": z, "image": img, "H": h, "W": w})

    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        """latents (B,4,h,w) bf16 or fp32, as the sampler leaves them (still multiplied by scaling_factor) ->
        image [B][8h][8w][3] float32 in about [-1, 1]."""
        B, _, h, w = latents.shape
        zb = latents.dtype == torch.bfloat16
        p = self._plan_for((B, h, w, zb), lambda a, z: self._build(B, h, w, a, z, zb))
        p.io["latents"].tensor.copy_(latents.reshape(B, 4, h * w) if zb else latents.float().reshape(B, 4, h * w))
        p.prog.run(torch.cuda.current_stream().cuda_stream)
        H, W = p.io["H"], p.io["W"]
        return p.io["image"].tensor.view(B, H, W, 4)[..., :3]

    @staticmethod
    def to_uint8(image: torch.Tensor) -> torch.Tensor:
        """(image / 2 + 0.5).clamp(0, 1) * 255, rounded (generate_images_sd1.py:169-171)."""
        return ((image / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8)
