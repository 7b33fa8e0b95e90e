"""Frozen UNet weights, repacked once at load time into the layouts the HIP kernels stream.

Input: a diffusers-format state dict (the keys of `unet/diffusion_pytorch_model.safetensors`, which is what
the reference loads at trainscripts/textsliders/model_util.py:67-72 / 169-174).  Output: bf16 device tensors
 - Linear            [N][K]                       (as stored)
 - Conv2d 3x3        [Cout][tap][Cin]  (tap = ky*3+kx)  -> implicit-GEMM K index tap*Cin + c
 - Conv2d 1x1        [Cout][Cin]
 - attn1 q/k/v       fused [3C][C];  attn2 k/v fused [2C][Dctx]
 - GEGLU proj        rows permuted into 64-row blocks [32 value rows | 32 gate rows] (fused GEGLU epilogue)
 - time_emb_proj     all ResnetBlock2D projections concatenated [sum(Cout)][temb] (one GEMV per UNet pass)
 - every matrix slh_gemm streams is then tile-packed (`pack_gemm_w`, slh_gemm_desc.w_layout = 1): [N/64][K/64]
   blocks of 64 rows x 64 k, 8 KB contiguous each, with the kernel's LDS swizzle already applied, so one
   workgroup's K step reads whole DRAM pages instead of 64 row fragments 2*K bytes apart
and, for the training pass only, the transposed / flipped copies that turn every backward-data product into
the same forward kernel (weights are frozen, so this is paid once): Linear W^T, conv3x3 [Cin][tap'][Cout]
with the taps flipped.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch

from .config import UNetConfig
from .modules import build_tree


def _conv3_pack(w: torch.Tensor) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def _geglu_perm(w: torch.Tensor) -> torch.Tensor:
    n2 = w.shape[0]
    n = n2 // 2
    a, g = w[:n], w[n:]
    rest = w.shape[1:]
    return torch.stack([a.reshape(n // 32, 32, *rest), g.reshape(n // 32, 32, *rest)], dim=1).reshape(n2, *rest).contiguous()


def _geglu_perm16(w: torch.Tensor) -> torch.Tensor:
    """Row order of slh_gemm_desc.geglu = 3: 32-row blocks [16 value rows | 16 gate rows] (the ping-pong tiles, whose waves own a
    number of 32-column blocks that need not be even)."""
    n2 = w.shape[0]
    n = n2 // 2
    a, g = w[:n], w[n:]
    rest = w.shape[1:]
    return torch.stack([a.reshape(n // 16, 16, *rest), g.reshape(n // 16, 16, *rest)], dim=1).reshape(n2, *rest).contiguous()


def pack_gemm_w(w: torch.Tensor) -> torch.Tensor:
    """[N][K] (K % 64 == 0) -> flat tile-packed layout of slh_gemm_desc.w_layout = 1 (include/sliders_hip.h):
    block (n>>6, k>>6) is 64 rows x 8 slots x 8 elements, slot s of row r stored at physical slot s ^ ((r>>1)&7);
    rows N..ceil64(N) are zero."""
    n, k = w.shape
    assert k % 64 == 0, k
    npad = (n + 63) // 64 * 64
    if npad != n:
        w = torch.cat([w, w.new_zeros(npad - n, k)], 0)
    t = w.reshape(npad // 64, 64, k // 64, 8, 8).permute(0, 2, 1, 3, 4)          # [nb][kb][r][slot][8]
    r = torch.arange(64, device=w.device)
    src = torch.arange(8, device=w.device)[None, :] ^ ((r >> 1) & 7)[:, None]     # physical slot p holds logical p ^ s(r)
    idx = src.view(1, 1, 64, 8, 1).expand(t.shape)
    return torch.gather(t, 3, idx).contiguous().reshape(-1)


def unpack_gemm_w(flat: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """Inverse of pack_gemm_w (tests, checkpoint export)."""
    npad = (n + 63) // 64 * 64
    t = flat.reshape(npad // 64, k // 64, 64, 8, 8)
    r = torch.arange(64, device=flat.device)
    src = torch.arange(8, device=flat.device)[None, :] ^ ((r >> 1) & 7)[:, None]
    t = torch.gather(t, 3, src.view(1, 1, 64, 8, 1).expand(t.shape))             # the swizzle is an involution
    return t.permute(0, 2, 1, 3, 4).reshape(npad, k)[:n].contiguous()


def fold_layernorm(w: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor, dtype=torch.bfloat16, device=None):
    """(W' = dtype(W * gamma), s = row sums of W' in fp32, b' = bias + W . beta in fp32) for slh_gemm_desc.ln_in:
    Linear(LayerNorm(x)) = rstd * (x . W'^T - mean * s) + b'.  s is taken from the ROUNDED W' the kernel multiplies with."""
    device = device if device is not None else w.device
    w32 = w.to(device=device, dtype=dtype).float()                      # the weights the unfused path multiplies with
    g32 = gamma.to(device=device, dtype=dtype).float()
    b32 = beta.to(device=device, dtype=dtype).float()
    wf = (w32 * g32[None, :]).to(dtype)
    s = wf.float().sum(1)
    bp = w32 @ b32
    if bias is not None:
        bp = bp + bias.to(device=device, dtype=dtype).float()
    return wf, s.contiguous(), bp.contiguous()


class WeightStore:
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], device, dtype=torch.bfloat16):
        self.cfg = cfg
        self.device = device
        self.dtype = dtype
        self.t: Dict[str, torch.Tensor] = {}
        self.gemm_shape: Dict[str, Tuple[int, int]] = {}
        # A/B switch for measurements only: SLIDERS_W_ROWMAJOR=1 keeps the GEMM matrices [N][K] (w_layout 0)
        self.packed = os.environ.get("SLIDERS_W_ROWMAJOR") is None
        self._sd = state_dict
        # LayerNorm folded into its consumer GEMMs in the no-grad passes (one more copy of the q|k|v, attn2.to_q and GEGLU
        # projection matrices, pre-scaled by the LayerNorm weight); SLIDERS_NO_LN_FOLD=1 keeps the LayerNorm launches
        self.ln_fold = os.environ.get("SLIDERS_NO_LN_FOLD") is None
        # GEGLU.proj also in the 16 | 16 block order (geglu = 3) for the no-grad passes: opens the 128 x 320 ping-pong tile to
        # ff.net.0.proj (2048 x 10240 x 1280: 512 tiles = two even rounds instead of 2.5 of the 128 x 128 tile): pass -0.6 ms in a
        # same-box A/B (profiles/r04_geglu16.txt).  Costs a second copy of those weights (+3.6 GB for SDXL); SLIDERS_GEGLU16=0
        # keeps only the 32 | 32 form
        self.geglu16 = os.environ.get("SLIDERS_GEGLU16", "1") == "1"
        self.temb_offsets: Dict[str, int] = {}
        self.resnet_paths: List[str] = []
        self._pack()
        self._dgrad_ready = False

    # -- helpers -----------------------------------------------------------------------------------
    def _put(self, name: str, t: torch.Tensor):
        self.t[name] = t.to(device=self.device, dtype=self.dtype).contiguous()

    def _put_gemm(self, name: str, t: torch.Tensor):
        """A matrix consumed by slh_gemm: stored tile-packed (shape kept in self.gemm_shape)."""
        t = t.to(device=self.device, dtype=self.dtype)
        self.gemm_shape[name] = tuple(t.shape)
        self.t[name] = pack_gemm_w(t) if self.packed else t.contiguous()

    def _put_ln_folded(self, wname: str, w: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor, perm=None, suffix: str = ""):
        """LayerNorm folded into the Linear that consumes it (slh_gemm_desc.ln_in; no-grad passes):
            Linear(LN(x)) = rstd * (x . W'^T - mean * s) + b',   W' = bf16(W * gamma),  s = row sums of W' (fp32, of the ROUNDED
            matrix the kernel multiplies with),  b' = bias + W . beta (fp32).
        Stored as <wname>.lnw (tile-packed like .w), .lns, .lnb; perm: the GEGLU row permutation of the fused epilogue."""
        wf, s, bp = fold_layernorm(w, bias, gamma, beta, self.dtype, self.device)
        if perm is not None:
            wf, s, bp = perm(wf), perm(s), perm(bp)
        self.gemm_shape[wname + ".lnw" + suffix] = tuple(wf.shape)
        self.t[wname + ".lnw" + suffix] = pack_gemm_w(wf) if self.packed else wf.contiguous()
        self.t[wname + ".lns" + suffix] = s.contiguous()
        self.t[wname + ".lnb" + suffix] = bp.contiguous()

    def gemm_matrix(self, name: str) -> torch.Tensor:
        """Row-major [N][K] view of a tile-packed matrix (copy)."""
        n, k = self.gemm_shape[name]
        return unpack_gemm_w(self.t[name], n, k) if self.packed else self.t[name]

    def ptr(self, name: str) -> int:
        return self.t[name].data_ptr()

    def has(self, name: str) -> bool:
        return name in self.t

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
                self._put_gemm(f"{name}.conv1.w", _conv3_pack(sd[f"{name}.conv1.weight"]))
                self._put(f"{name}.conv1.b", sd[f"{name}.conv1.bias"])
                self._put(f"{name}.norm2.g", sd[f"{name}.norm2.weight"])
                self._put(f"{name}.norm2.b", sd[f"{name}.norm2.bias"])
                self._put_gemm(f"{name}.conv2.w", _conv3_pack(sd[f"{name}.conv2.weight"]))
                self._put(f"{name}.conv2.b", sd[f"{name}.conv2.bias"])
                if f"{name}.conv_shortcut.weight" in sd:
                    w = sd[f"{name}.conv_shortcut.weight"]
                    self._put_gemm(f"{name}.conv_shortcut.w", w.reshape(w.shape[0], w.shape[1]))
                    self._put(f"{name}.conv_shortcut.b", sd[f"{name}.conv_shortcut.bias"])
                self.temb_offsets[name] = off
                temb_w.append(sd[f"{name}.time_emb_proj.weight"])
                temb_b.append(sd[f"{name}.time_emb_proj.bias"])
                off += node.out_dim
            elif node.cls in ("Downsample2D", "Upsample2D"):
                self._put_gemm(f"{name}.conv.w", _conv3_pack(sd[f"{name}.conv.weight"]))
                self._put(f"{name}.conv.b", sd[f"{name}.conv.bias"])
            elif node.cls == "Transformer2DModel":
                self._put(f"{name}.norm.g", sd[f"{name}.norm.weight"])
                self._put(f"{name}.norm.b", sd[f"{name}.norm.bias"])
                for pj in ("proj_in", "proj_out"):
                    w = sd[f"{name}.{pj}.weight"]
                    self._put_gemm(f"{name}.{pj}.w", w.reshape(w.shape[0], w.shape[1]))
                    self._put(f"{name}.{pj}.b", sd[f"{name}.{pj}.bias"])
            elif node.cls == "BasicTransformerBlock":
                for nm in ("norm1", "norm2", "norm3"):
                    self._put(f"{name}.{nm}.g", sd[f"{name}.{nm}.weight"])
                    self._put(f"{name}.{nm}.b", sd[f"{name}.{nm}.bias"])
                a1, a2 = f"{name}.attn1", f"{name}.attn2"
                self._put_gemm(f"{a1}.qkv.w", torch.cat([sd[f"{a1}.to_q.weight"], sd[f"{a1}.to_k.weight"],
                                                    sd[f"{a1}.to_v.weight"]], 0))
                self._put_gemm(f"{a1}.out.w", sd[f"{a1}.to_out.0.weight"])
                self._put(f"{a1}.out.b", sd[f"{a1}.to_out.0.bias"])
                self._put_gemm(f"{a2}.q.w", sd[f"{a2}.to_q.weight"])
                self._put_gemm(f"{a2}.kv.w", torch.cat([sd[f"{a2}.to_k.weight"], sd[f"{a2}.to_v.weight"]], 0))
                self._put_gemm(f"{a2}.out.w", sd[f"{a2}.to_out.0.weight"])
                self._put(f"{a2}.out.b", sd[f"{a2}.to_out.0.bias"])
                self._put_gemm(f"{name}.ff1.w", _geglu_perm(sd[f"{name}.ff.net.0.proj.weight"]))
                self._put(f"{name}.ff1.b", _geglu_perm(sd[f"{name}.ff.net.0.proj.bias"]))
                if self.geglu16:
                    # second copy in the 16 | 16 block order of slh_gemm_desc.geglu = 3 (the no-grad passes: lets the tuner give
                    # GEGLU.proj tiles whose waves own an odd number of 32-column blocks, e.g. 256 x 320); the training forward
                    # keeps the 32 | 32 copy - its backward reads geglu_pre in that order
                    self._put_gemm(f"{name}.ff1.w16", _geglu_perm16(sd[f"{name}.ff.net.0.proj.weight"]))
                    self._put(f"{name}.ff1.b16", _geglu_perm16(sd[f"{name}.ff.net.0.proj.bias"]))
                self._put_gemm(f"{name}.ff2.w", sd[f"{name}.ff.net.2.weight"])
                self._put(f"{name}.ff2.b", sd[f"{name}.ff.net.2.bias"])
                if self.ln_fold:
                    n1, n2, n3 = (( sd[f"{name}.{nm}.weight"], sd[f"{name}.{nm}.bias"]) for nm in ("norm1", "norm2", "norm3"))
                    self._put_ln_folded(f"{a1}.qkv", torch.cat([sd[f"{a1}.to_q.weight"], sd[f"{a1}.to_k.weight"],
                                                                sd[f"{a1}.to_v.weight"]], 0), None, *n1)
                    self._put_ln_folded(f"{a2}.q", sd[f"{a2}.to_q.weight"], None, *n2)
                    self._put_ln_folded(f"{name}.ff1", sd[f"{name}.ff.net.0.proj.weight"], sd[f"{name}.ff.net.0.proj.bias"], *n3,
                                        perm=_geglu_perm)
                    if self.geglu16:
                        self._put_ln_folded(f"{name}.ff1", sd[f"{name}.ff.net.0.proj.weight"], sd[f"{name}.ff.net.0.proj.bias"], *n3,
                                            perm=_geglu_perm16, suffix="16")
        # all cross-attention K/V projections read the SAME text embeddings: one [sum(C) K rows | sum(C) V rows][Dctx]
        # matrix lets a UNet pass compute them in one full-chip launch instead of one 120-workgroup launch per
        # transformer block, and transposes every V with one more (tile-packed blocks are row-block major, so packed
        # sub-matrices simply concatenate)
        self.kv_all_offset: Dict[str, Tuple[int, int]] = {}      # attn2 path -> (first K column, first V column)
        self.kv_all_vbase = 0
        kv_names = [n for n in self.t if n.endswith(".attn2.kv.w")]
        if kv_names and self.packed and all(self.gemm_shape[n][0] % 128 == 0 for n in kv_names):
            kdim = self.gemm_shape[kv_names[0]][1]
            halves = [self.gemm_shape[n][0] // 2 for n in kv_names]
            self.kv_all_vbase = sum(halves)
            kparts, vparts, row = [], [], 0
            for n, c in zip(kv_names, halves):
                flat = self.t[n]
                kparts.append(flat[: c * kdim])
                vparts.append(flat[c * kdim:])
                self.kv_all_offset[n[:-len(".kv.w")]] = (row, self.kv_all_vbase + row)
                row += c
            self.t["attn2_kv_all.w"] = torch.cat(kparts + vparts)
            self.gemm_shape["attn2_kv_all.w"] = (2 * self.kv_all_vbase, kdim)
        self.temb_total = off
        self._put("temb_proj.w", torch.cat(temb_w, 0))
        self._put("temb_proj.b", torch.cat(temb_b, 0))

    # -- backward-data layouts (training pass only) -------------------------------------------------
    def ensure_dgrad(self):
        if self._dgrad_ready:
            return
        root = build_tree(self.cfg)

        def dg(key):  # packed [Cout][tap][Cin] -> [Cin][flipped tap][Cout]
            w = self.gemm_matrix(key)
            co = w.shape[0]
            ci = w.shape[1] // 9
            return w.view(co, 3, 3, ci).flip(1, 2).permute(3, 1, 2, 0).reshape(ci, 9 * co)

        for name, node in root.named_modules():
            if node.cls == "ResnetBlock2D":
                self._put_gemm(f"{name}.conv1.wT", dg(f"{name}.conv1.w"))
                self._put_gemm(f"{name}.conv2.wT", dg(f"{name}.conv2.w"))
                if self.has(f"{name}.conv_shortcut.w"):
                    self._put_gemm(f"{name}.conv_shortcut.wT", self.gemm_matrix(f"{name}.conv_shortcut.w").t())
            elif node.cls in ("Downsample2D", "Upsample2D"):
                self._put_gemm(f"{name}.conv.wT", dg(f"{name}.conv.w"))
            elif node.cls == "Transformer2DModel":
                for pj in ("proj_in", "proj_out"):
                    self._put_gemm(f"{name}.{pj}.wT", self.gemm_matrix(f"{name}.{pj}.w").t())
            elif node.cls == "BasicTransformerBlock":
                for k in ("attn1.qkv", "attn1.out", "attn2.q", "attn2.out", "ff1", "ff2"):
                    self._put_gemm(f"{name}.{k}.wT", self.gemm_matrix(f"{name}.{k}.w").t())
        self._dgrad_ready = True

    def release_source(self):
        """Drop the reference to the source state dict (host memory)."""
        self._sd = None

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())
