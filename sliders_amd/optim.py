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


class Prodigy(torch.optim.Optimizer):
    """`train.optimizer: prodigy` (train_util.py:369-372 returns prodigyopt.Prodigy; requirements.txt pins prodigyopt==1.0).
    The package is absent here (parity unpinned): this follows that release's published algorithm (Mishchenko & Defazio
    2023, "Prodigy: An Expeditiously Adaptive Parameter-Free Learner", Alg. 4 with the package's d/d0 rescaling, beta3 =
    sqrt(beta2) default, optional bias correction, safeguard warm-up, decoupled weight decay), same constructor arguments
    and defaults, state tensors in the parameter dtype like the package's zeros_like(p).

    Written as tensor ops on the parameter's own device (the adapter parameters are one flat ~4M-element bf16 buffer: a
    dozen elementwise launches and two reductions per optimizer step, once per ~28 UNet passes - not kernel material).
    The two global reductions that drive the step-size estimate accumulate in fp32 (`torch.dot` / `sum` of `.float()`
    views; the package reduces in the parameter dtype, i.e. with an 8-bit mantissa for bf16 parameters)."""

    def __init__(self, params, lr: float = 1.0, betas=(0.9, 0.999), beta3=None, eps: float = 1e-8, weight_decay: float = 0.0,
                 decouple: bool = True, use_bias_correction: bool = False, safeguard_warmup: bool = False, d0: float = 1e-6,
                 d_coef: float = 1.0, growth_rate: float = float("inf")):
        if not 0.0 < d0:
            raise ValueError(f"Invalid d0 value: {d0}")
        if not 0.0 < lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 < eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid beta parameters: {betas}")
        if decouple and weight_decay > 0:
            print("Using decoupled weight decay")
        super().__init__(params, dict(lr=lr, betas=betas, beta3=beta3, eps=eps, weight_decay=weight_decay, d=d0, d0=d0,
                                      d_max=d0, d_numerator=0.0, d_coef=d_coef, k=0, growth_rate=growth_rate,
                                      use_bias_correction=use_bias_correction, decouple=decouple,
                                      safeguard_warmup=safeguard_warmup))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g0 = self.param_groups[0]
        beta1, beta2 = g0["betas"]
        beta3 = g0["beta3"] if g0["beta3"] is not None else beta2 ** 0.5
        k, d, d_max, d_coef, growth_rate = g0["k"], g0["d"], g0["d_max"], g0["d_coef"], g0["growth_rate"]
        lr = max(g["lr"] for g in self.param_groups)
        bias_correction = ((1 - beta2 ** (k + 1)) ** 0.5) / (1 - beta1 ** (k + 1)) if g0["use_bias_correction"] else 1.0
        dlr = d * lr * bias_correction
        d_numerator = g0["d_numerator"] * beta3
        d_denom = 0.0
        for group in self.param_groups:
            decay, d0, group_lr = group["weight_decay"], group["d0"], group["lr"]
            if group_lr not in (lr, 0.0):
                raise RuntimeError("Setting different lr values in different parameter groups is only supported for values of 0")
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                if decay != 0 and not group["decouple"]:
                    grad.add_(p, alpha=decay)
                st = self.state[p]
                if "step" not in st:
                    st["step"] = 0
                    st["s"] = torch.zeros_like(p)
                    st["p0"] = p.detach().clone()
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                if group_lr > 0.0:
                    # d / d0 instead of d keeps the running sums away from underflow
                    d_numerator += (d / d0) * dlr * torch.dot(grad.flatten().float(), (st["p0"] - p).flatten().float()).item()
                    st["exp_avg"].mul_(beta1).add_(grad, alpha=d * (1 - beta1))
                    st["exp_avg_sq"].mul_(beta2).addcmul_(grad, grad, value=d * d * (1 - beta2))
                    st["s"].mul_(beta3).add_(grad, alpha=(d / d0) * (d if group["safeguard_warmup"] else dlr))
                    d_denom += st["s"].float().abs().sum().item()
        d_hat = d
        if d_denom == 0:            # no gradient seen: nothing to do
            return loss
        if lr > 0.0:
            d_hat = d_coef * d_numerator / d_denom
            if d == g0["d0"]:
                d = max(d, d_hat)
            d_max = max(d_max, d_hat)
            d = min(d_max, d * growth_rate)
        for group in self.param_groups:
            group["d_numerator"], group["d_denom"] = d_numerator, d_denom
            group["d"], group["d_max"], group["d_hat"] = d, d_max, d_hat
            decay, eps = group["weight_decay"], group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                st["step"] += 1
                denom = st["exp_avg_sq"].sqrt().add_(d * eps)
                if decay != 0 and group["decouple"]:
                    p.add_(p, alpha=-decay * dlr)
                p.addcdiv_(st["exp_avg"], denom, value=-dlr)
            group["k"] = k + 1
        return loss
