"""UNetEngine: the drop-in for the `unet` object the reference passes to predict_noise[_xl]
(trainscripts/textsliders/train_util.py:145-171, 220-260):

    unet(latent_model_input, timestep, encoder_hidden_states=E[, added_cond_kwargs={...}]).sample

Same call signature, same (B,4,H,W) latent layout at the boundary, same attributes the reference's callers
touch (.to/.eval/.requires_grad_/.enable_xformers_memory_efficient_attention/.config.in_channels/.in_channels/
.dtype).  Everything between the input copy and the output tensor is ONE slh_run_program call on the current
PyTorch-ROCm stream; PyTorch only owns the memory.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import lib
from .arena import Arena
from .config import UNetConfig
from .lora_store import LoraStore
from .planner import BackwardPlan, UNetPlan
from .weights import WeightStore


class UNetOutput:
    def __init__(self, sample: torch.Tensor):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class UNetEngine:
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0",
                 arena_bytes: Optional[int] = None, ctx_len: int = 77, zarena_bytes: Optional[int] = None):
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("UNetEngine needs a ROCm GPU (there is no CPU fallback for the hot path)")
        lib.load()
        self.weights = WeightStore(cfg, state_dict, self.device)
        self.ctx_len = ctx_len
        self.config = _Cfg(in_channels=cfg.in_channels, sample_size=cfg.sample_size)
        self.in_channels = cfg.in_channels
        self.dtype = torch.bfloat16
        self.lora: Optional[LoraStore] = None
        self.lora_active = False
        self.lora_scale = torch.zeros(1, dtype=torch.float32, device=self.device)   # multiplier * alpha / rank, read by the kernels
        self.lora_scale_host = 0.0
        self._plans: Dict[Tuple, UNetPlan] = {}
        self._arena_bytes = arena_bytes
        self.arena: Optional[Arena] = None
        self.arena_off: Optional[Arena] = None     # adapter-free plans live apart: the trainer runs its frozen pass on a second stream
        # zero-initialised words: the arrival tickets of the fixed-order GroupNorm reductions, the time-embedding adapter's
        # column sums (~150 KB per cached (B, H, W) shape; GroupNorm statistics and split-K slabs need no zeroing any more).
        # Plans never give their span back; when the arena runs low every cached plan is dropped and it restarts from 0.
        self.zarena = Arena(int(zarena_bytes or (64 << 20)), self.device, "zero-init accumulators")
        self.plan_flushes = 0
        self.train_plan: Optional[UNetPlan] = None
        self.one = torch.ones(1, dtype=torch.float32, device=self.device)
        self.grad_all_samples = False
        self.autograd_param: Optional[torch.nn.Parameter] = None   # set by LoRANetwork.flat_parameter()

    # ---- reference-facing no-ops ----------------------------------------------------------------
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def enable_xformers_memory_efficient_attention(self):
        return None

    # ---- LoRA switch (driven by sliders_amd.lora.LoRANetwork) ------------------------------------
    def attach_lora(self, store: LoraStore):
        self.lora = store
        self._plans = {k: v for k, v in self._plans.items() if k[3] == "off"}

    def set_lora(self, active: bool, multiplier: float = 1.0):
        self.lora_active = bool(active) and self.lora is not None
        if self.lora is not None:
            self.lora_scale_host = float(multiplier) * self.lora.scale if active else 0.0
            self.lora_scale.fill_(self.lora_scale_host)

    # ---- planning ---------------------------------------------------------------------------------
    def _virtual_size(self, B, H, W, mode) -> int:
        va = Arena(1 << 50, None, "virtual")
        vz = Arena(1 << 40, None, "virtualz")
        p = UNetPlan(self.cfg, _VirtualWeights(self.weights), va, vz, B, H, W, self.ctx_len,
                     _VirtualLora(self.lora) if mode != "off" else None, mode, 0)
        if mode == "train":
            b0, nb = self._grad_samples(B)
            BackwardPlan(p, b0, nb, 0)
        return va.high_water

    def _grad_samples(self, B: int):
        """Samples that receive a non-zero gradient: the text half of the CFG pair (see BackwardPlan)."""
        if self.grad_all_samples or B == 1:
            return 0, B
        return B // 2, B // 2

    def _ensure_arena(self, need: int, off: bool = False):
        cur = self.arena_off if off else self.arena
        if cur is not None and cur.capacity >= need:
            return
        # grow: every cached plan of that arena holds raw pointers into the old one, so they are rebuilt lazily
        self._plans = {k: v for k, v in self._plans.items() if (k[3] == "off") != off}
        if not off:
            self.train_plan = None
        torch.cuda.synchronize()
        if off:
            self.arena_off = None
            self.arena_off = Arena(need + (1 << 20), self.device, "activations (adapter-free plans)")
        else:
            self.arena = None
            self.arena = Arena(max(need, self._arena_bytes or 0) + (1 << 20), self.device, "activations")

    def plan(self, B: int, H: int, W: int, mode: str) -> UNetPlan:
        key = (B, H, W, mode)
        p = self._plans.get(key)
        if p is not None:
            return p
        # dynamic_resolution / per-prompt resolutions reach dozens of shapes (16 buckets on SD-1.x, 64 on SDXL): make room
        # in the zero-init arena before it overflows (the plans of one shape - on, train, off - take Z_PER_SHAPE at most)
        if self.zarena.capacity - self.zarena.mark() < min(self.Z_PER_SHAPE, self.zarena.capacity // 2):
            self.flush_plans()
        # size the shared arena for the requested plan and, once adapters are attached, for the training plan
        # of the same shape (so 'on' -> 'train' does not trigger a regrow that invalidates cached plans)
        modes = {mode} | ({"train"} if (self.lora is not None and mode != "off") else set())
        need = max(self._virtual_size(B, H, W, m) for m in modes)
        off = mode == "off"
        self._ensure_arena(need, off)
        if mode == "train":
            self.weights.ensure_dgrad()
        # the plans of one arena share it from offset 0 (they never run concurrently); adapter-free plans have their own
        # arena so that the frozen predictions of an iteration can overlap the training forward on a second stream
        arena = self.arena_off if off else self.arena
        arena.reset(0)
        p = UNetPlan(self.cfg, self.weights, arena, self.zarena, B, H, W, self.ctx_len,
                     self.lora if mode != "off" else None, mode, self.lora_scale.data_ptr())
        if mode == "train":
            b0, nb = self._grad_samples(B)
            p.backward = BackwardPlan(p, b0, nb, self.one.data_ptr())
        p.arena_end = arena.mark()
        self._plans[key] = p
        return p

    Z_PER_SHAPE = 2 << 20

    def flush_plans(self):
        """Drop every cached plan (they hold raw pointers into the arenas) and restart the zero-init arena."""
        torch.cuda.synchronize()
        self._plans.clear()
        self.train_plan = None
        self.zarena.reset(0)
        self.plan_flushes += 1

    def run_backward(self, p: Optional[UNetPlan] = None, d_eps: Optional[torch.Tensor] = None):
        """Backward of the last train-mode forward.  d_eps: gradient w.r.t. the returned epsilon for the
        gradient-carrying samples, (nb,4,H,W); None if the caller already filled backward.deps_pix
        (slh_guidance_loss writes it directly).  Accumulates into lora.grads (fp32)."""
        p = p or self.train_plan
        bw = p.backward
        if d_eps is not None:
            nb, C, H, W = d_eps.shape
            bw.deps_pix.tensor.copy_(d_eps.to(torch.bfloat16).float().permute(0, 2, 3, 1).reshape(nb * H * W, C))
        bw.prog.run(torch.cuda.current_stream().cuda_stream)

    # ---- the model call ---------------------------------------------------------------------------
    def _mode(self) -> str:
        if not self.lora_active:
            return "off"
        return "train" if torch.is_grad_enabled() else "on"

    def load_inputs(self, p: UNetPlan, sample, timestep, encoder_hidden_states, added_cond_kwargs):
        B = p.B
        io = p.io
        io["sample"].tensor.copy_(sample.to(torch.bfloat16))
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor(float(t))
        io["t"].tensor.copy_(t.to(device=self.device, dtype=torch.float32).reshape(-1, 1).expand(B, 1))
        io["ctx"].tensor.copy_(encoder_hidden_states.to(torch.bfloat16))
        if self.cfg.is_xl:
            te = added_cond_kwargs["text_embeds"]
            ti = added_cond_kwargs["time_ids"]
            io["time_ids"].tensor.copy_(ti.to(torch.float32).reshape(B, 6))
            io["add_in"].tensor[:, : self.cfg.pooled_dim].copy_(te.to(torch.bfloat16))

    def forward_plan(self, p: UNetPlan):
        p.prog.run(torch.cuda.current_stream().cuda_stream)

    def __call__(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None,
                 return_dict: bool = True, mode: Optional[str] = None):
        B, _, H, W = sample.shape
        mode = mode or self._mode()
        p = self.plan(B, H, W, mode)
        self.load_inputs(p, sample, timestep, encoder_hidden_states, added_cond_kwargs)
        self.forward_plan(p)
        if mode == "train":
            self.train_plan = p
        out = p.io["eps"].tensor.clone()
        if mode == "train" and torch.is_grad_enabled() and self.autograd_param is not None:
            out = _EpsBridge.apply(self.autograd_param, out, self, p)
        if not return_dict:
            return (out,)
        return UNetOutput(out)

    forward = __call__


class _EpsBridge(torch.autograd.Function):
    """Lets the reference-shaped loop call `loss.backward()`: the engine output becomes a function of the flat
    LoRA Parameter; backward replays the HIP backward program and returns the packed gradient (bf16, like
    the reference's param.grad).  Samples outside the gradient-carrying half must receive a zero gradient
    (true for the reference's guidance_scale=1 target prediction); otherwise set engine.grad_all_samples."""

    @staticmethod
    def forward(ctx, flat_param, eps, engine, plan):
        ctx.engine, ctx.plan = engine, plan
        # the adapter scale is an input of the backward kernels too (dA, dB carry a factor multiplier*alpha/rank); the
        # reference calls loss.backward() AFTER leaving `with network:` (train_lora_xl.py:302-345), where the live
        # multiplier is 0 again, so the value of the forward is recorded here like autograd records it
        ctx.scale = engine.lora_scale_host
        return eps.view_as(eps)

    @staticmethod
    def backward(ctx, g):
        eng, p = ctx.engine, ctx.plan
        bw = p.backward
        b0, nb = bw.b0, bw.nb
        if b0 > 0 and bool((g[:b0] != 0).any()):
            raise RuntimeError("non-zero gradient on the unconditional half: build the engine plan with "
                               "engine.grad_all_samples = True")
        eng.lora.grads.zero_()
        live = eng.lora_scale_host
        eng.lora_scale.fill_(ctx.scale)
        eng.run_backward(p, d_eps=g[b0:b0 + nb])
        eng.lora_scale.fill_(live)
        return eng.lora.grads.to(torch.bfloat16), None, None, None


class _VirtualWeights:
    """Pointer-less stand-in so the planner can be dry-run for sizing."""

    def __init__(self, w: WeightStore):
        self._w = w
        self.temb_offsets = w.temb_offsets
        self.temb_total = w.temb_total
        self.resnet_paths = w.resnet_paths
        self.packed = w.packed
        self.kv_all_offset = w.kv_all_offset
        self.kv_all_vbase = w.kv_all_vbase
        self.gemm_shape = w.gemm_shape
        self.ln_fold = w.ln_fold
        self.geglu16 = getattr(w, "geglu16", False)    # the dry run must plan GEGLU.proj on the tile the real plan will take

    def ptr(self, name):
        return 0x1000

    def has(self, name):
        return self._w.has(name)


class _VirtualLora:
    def __init__(self, s: LoraStore):
        self._s = s
        self.temb_entries = s.temb_entries
        self.temb_down_off = s.temb_down_off
        self.temb_up_off = s.temb_up_off
        self.temb_tcol = _FakeTensor()
        self.params = _FakeTensor()

    def fused_group(self, paths):
        return self._s.fused_group(paths)

    def up_t_offset(self, grp):
        return self._s.up_t_offset(grp)

    def down_ptr(self, e):
        return 0x2000

    def up_ptr(self, e):
        return 0x2000

    def gdown_ptr(self, e):
        return 0x2000

    def gup_ptr(self, e):
        return 0x2000


class _FakeTensor:
    device = "cpu"

    def data_ptr(self):
        return 0x3000
