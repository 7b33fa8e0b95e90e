"""torch.optim-shaped front ends of the fused flat optimizer kernels, for the reference-shaped loop of INTEGRATION.md
section 1 (`optimizer_module = train_util.get_optimizer(name); optimizer = optimizer_module(params, lr=..., **kwargs)`,
trainscripts/textsliders/train_lora_xl.py:92-103).  The fused trainer (sliders_amd/trainer.py) calls the kernels directly."""
from __future__ import annotations

import torch

from . import lib


class Lion(torch.optim.Optimizer):
    """lion_pytorch.Lion (requirements.txt:5: lion_pytorch==0.1.2) over slh_lion: same constructor arguments and defaults,
    one bf16 moment per parameter, the package's op order and bf16 rounding points (see include/sliders_hip.h)."""

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.99), weight_decay: float = 0.0):
        if lr <= 0.0 or not all(0.0 <= b <= 1.0 for b in betas):
            raise ValueError("Lion: lr must be positive and betas within [0, 1]")
        super().__init__(params, dict(lr=lr, betas=betas, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.bfloat16 and p.is_contiguous()):
                    raise RuntimeError("sliders_amd.optim.Lion updates contiguous bf16 parameters on the GPU (slh_lion); there is "
                                       "no CPU fallback")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p)
                g32 = p.grad.to(torch.float32).contiguous()        # the kernel re-rounds to bf16: exact for a bf16 .grad
                d = lib.LionDesc(param=p.data_ptr(), exp_avg=st["exp_avg"].data_ptr(), grad=g32.data_ptr(), n=p.numel(),
                                 lr=group["lr"], beta1=group["betas"][0], beta2=group["betas"][1],
                                 weight_decay=group["weight_decay"], grad_scale=1.0)
                lib.call(lib.OP_LION, d, torch.cuda.current_stream().cuda_stream)
        return loss
