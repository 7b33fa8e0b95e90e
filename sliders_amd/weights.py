"""Frozen UNet weights, repacked once at load time into the layouts the HIP kernels stream.

Input: a diffusers-format state dict (the keys of `unet/diffusion_pytorch_model.safetensors`, which is what
the reference loads at trainscripts/textsliders/model_util.py:67-72 / 169-174).  Output: bf16 device tensors
 - Linear            [N][K]                       (as stored)
 - Conv2d 3x3        [Cout][tap][Cin]  (tap = ky*3+kx)  -> implicit-GEMM K index tap*Cin + c
 - Conv2d 1x1        [Cout][Cin]
 - attn1 q/k/v       fused [3C][C];  attn2 k/v fused [2C][Dctx]
 - GEGLU proj        rows permuted into 64-row blocks [32 value rows | 32 gate rows] (fused GEGLU epilogue)
 - time_emb_proj     all ResnetBlock2D projections concatenated [sum(Cout)][temb] (one GEMV per UNet pass)
and, for the training pass only, the transposed / flipped copies that turn every backward-data product into
the same forward kernel (weights are frozen, so this is paid once): Linear W^T, conv3x3 [Cin][tap'][Cout]
with the taps flipped.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .config import UNetConfig
from .modules import build_tree


def _conv3_pack(w: torch.Tensor) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def _conv3_dgrad_pack(w: torch.Tensor) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, kh * kw * co).contiguous()


def _geglu_perm(w: torch.Tensor) -> torch.Tensor:
    n2 = w.shape[0]
    n = n2 // 2
    a, g = w[:n], w[n:]
    rest = w.shape[1:]
    return torch.stack([a.reshape(n // 32, 32, *rest), g.reshape(n // 32, 32, *rest)], dim=1).reshape(n2, *rest).contiguous()


class WeightStore:
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], device, dtype=torch.bfloat16):
        self.cfg = cfg
        self.device = device
        self.dtype = dtype
        self.t: Dict[str, torch.Tensor] = {}
        self._sd = state_dict
        self.temb_offsets: Dict[str, int] = {}
        self.resnet_paths: List[str] = []
        self._pack()
        self._dgrad_ready = False

    # -- helpers -----------------------------------------------------------------------------------
    def _put(self, name: str, t: torch.Tensor):
        self.t[name] = t.to(device=self.device, dtype=self.dtype).contiguous()

    def ptr(self, name: str) -> int:
        return self.t[name].data_ptr()

    def has(self, name: str) -> bool:
        return name in self.t

    def _w(self, key: str) -> torch.Tensor:
        return self._sd[key]

    # -- forward layouts ---------------------------------------------------------------------------
    def _pack(self):
        cfg, sd = self.cfg, self._sd
        root = build_tree(cfg)
        self._put("conv_in.w", _conv3_pack(sd["conv_in.weight"]))
        self._put("conv_in.b", sd["conv_in.bias"])
        for emb in ("time_embedding",) + (("add_embedding",) if cfg.is_xl else ()):
            for lin in ("linear_1", "linear_2"):
                self._put(f"{emb}.{lin}.w", sd[f"{emb}.{lin}.weight"])
                self._put(f"{emb}.{lin}.b", sd[f"{emb}.{lin}.bias"])
        self._put("conv_norm_out.g", sd["conv_norm_out.weight"])
        self._put("conv_norm_out.b", sd["conv_norm_out.bias"])
        self._put("conv_out.w", _conv3_pack(sd["conv_out.weight"]))
        self._put("conv_out.b", sd["conv_out.bias"])

        temb_w, temb_b, off = [], [], 0
        for name, node in root.named_modules():
            if node.cls == "ResnetBlock2D":
                self.resnet_paths.append(name)
                self._put(f"{name}.norm1.g", sd[f"{name}.norm1.weight"])
                self._put(f"{name}.norm1.b", sd[f"{name}.norm1.bias"])
                self._put(f"{name}.conv1.w", _conv3_pack(sd[f"{name}.conv1.weight"]))
                self._put(f"{name}.conv1.b", sd[f"{name}.conv1.bias"])
                self._put(f"{name}.norm2.g", sd[f"{name}.norm2.weight"])
                self._put(f"{name}.norm2.b", sd[f"{name}.norm2.bias"])
                self._put(f"{name}.conv2.w", _conv3_pack(sd[f"{name}.conv2.weight"]))
                self._put(f"{name}.conv2.b", sd[f"{name}.conv2.bias"])
                if f"{name}.conv_shortcut.weight" in sd:
                    w = sd[f"{name}.conv_shortcut.weight"]
                    self._put(f"{name}.conv_shortcut.w", w.reshape(w.shape[0], w.shape[1]))
                    self._put(f"{name}.conv_shortcut.b", sd[f"{name}.conv_shortcut.bias"])
                self.temb_offsets[name] = off
                temb_w.append(sd[f"{name}.time_emb_proj.weight"])
                temb_b.append(sd[f"{name}.time_emb_proj.bias"])
                off += node.out_dim
            elif node.cls in ("Downsample2D", "Upsample2D"):
                self._put(f"{name}.conv.w", _conv3_pack(sd[f"{name}.conv.weight"]))
                self._put(f"{name}.conv.b", sd[f"{name}.conv.bias"])
            elif node.cls == "Transformer2DModel":
                self._put(f"{name}.norm.g", sd[f"{name}.norm.weight"])
                self._put(f"{name}.norm.b", sd[f"{name}.norm.bias"])
                for pj in ("proj_in", "proj_out"):
                    w = sd[f"{name}.{pj}.weight"]
                    self._put(f"{name}.{pj}.w", w.reshape(w.shape[0], w.shape[1]))
                    self._put(f"{name}.{pj}.b", sd[f"{name}.{pj}.bias"])
            elif node.cls == "BasicTransformerBlock":
                for nm in ("norm1", "norm2", "norm3"):
                    self._put(f"{name}.{nm}.g", sd[f"{name}.{nm}.weight"])
                    self._put(f"{name}.{nm}.b", sd[f"{name}.{nm}.bias"])
                a1, a2 = f"{name}.attn1", f"{name}.attn2"
                self._put(f"{a1}.qkv.w", torch.cat([sd[f"{a1}.to_q.weight"], sd[f"{a1}.to_k.weight"],
                                                    sd[f"{a1}.to_v.weight"]], 0))
                self._put(f"{a1}.out.w", sd[f"{a1}.to_out.0.weight"])
                self._put(f"{a1}.out.b", sd[f"{a1}.to_out.0.bias"])
                self._put(f"{a2}.q.w", sd[f"{a2}.to_q.weight"])
                self._put(f"{a2}.kv.w", torch.cat([sd[f"{a2}.to_k.weight"], sd[f"{a2}.to_v.weight"]], 0))
                self._put(f"{a2}.out.w", sd[f"{a2}.to_out.0.weight"])
                self._put(f"{a2}.out.b", sd[f"{a2}.to_out.0.bias"])
                self._put(f"{name}.ff1.w", _geglu_perm(sd[f"{name}.ff.net.0.proj.weight"]))
                self._put(f"{name}.ff1.b", _geglu_perm(sd[f"{name}.ff.net.0.proj.bias"]))
                self._put(f"{name}.ff2.w", sd[f"{name}.ff.net.2.weight"])
                self._put(f"{name}.ff2.b", sd[f"{name}.ff.net.2.bias"])
        self.temb_total = off
        self._put("temb_proj.w", torch.cat(temb_w, 0))
        self._put("temb_proj.b", torch.cat(temb_b, 0))

    # -- backward-data layouts (training pass only) -------------------------------------------------
    def ensure_dgrad(self):
        if self._dgrad_ready:
            return
        root = build_tree(self.cfg)

        def dg(key):  # packed [Cout][tap][Cin] -> [Cin][flipped tap][Cout]
            w = self.t[key]
            co = w.shape[0]
            ci = w.shape[1] // 9
            return w.view(co, 3, 3, ci).flip(1, 2).permute(3, 1, 2, 0).reshape(ci, 9 * co)

        for name, node in root.named_modules():
            if node.cls == "ResnetBlock2D":
                self._put(f"{name}.conv1.wT", dg(f"{name}.conv1.w"))
                self._put(f"{name}.conv2.wT", dg(f"{name}.conv2.w"))
                if self.has(f"{name}.conv_shortcut.w"):
                    self._put(f"{name}.conv_shortcut.wT", self.t[f"{name}.conv_shortcut.w"].t())
            elif node.cls in ("Downsample2D", "Upsample2D"):
                self._put(f"{name}.conv.wT", dg(f"{name}.conv.w"))
            elif node.cls == "Transformer2DModel":
                for pj in ("proj_in", "proj_out"):
                    self._put(f"{name}.{pj}.wT", self.t[f"{name}.{pj}.w"].t())
            elif node.cls == "BasicTransformerBlock":
                for k in ("attn1.qkv", "attn1.out", "attn2.q", "attn2.out", "ff1", "ff2"):
                    self._put(f"{name}.{k}.wT", self.t[f"{name}.{k}.w"].t())
        self._dgrad_ready = True

    def release_source(self):
        """Drop the reference to the source state dict (host memory)."""
        self._sd = None

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())
