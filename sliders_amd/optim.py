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


class _DAdaptBase(torch.optim.Optimizer):
    """What DAdaptAdam and DAdaptLion of dadaptation 3.1 share: ONE step-size estimate d for all parameter groups kept in
    group 0 and mirrored into the others, group learning rates restricted to {lr, 0}, the two global reductions
    <direction, s> and ||s||_1 accumulated here in at least fp32 whatever the state dtype (the package reduces in the parameter dtype,
    an 8-bit mantissa for bf16 parameters)."""

    @staticmethod
    def _dot(a, b) -> float:
        acc = torch.promote_types(a.dtype, torch.float32)       # at least fp32, float64 states reduce in float64
        return torch.dot(a.flatten().to(acc), b.flatten().to(acc)).item()

    @staticmethod
    def _l1(s) -> float:
        return s.to(torch.promote_types(s.dtype, torch.float32)).abs().sum().item()

    def _common_lr(self) -> float:
        lr = max(g["lr"] for g in self.param_groups)
        for g in self.param_groups:
            if g["lr"] not in (lr, 0.0):
                raise RuntimeError("Setting different lr values in different parameter groups is only supported for values of 0")
        return lr


class DAdaptAdam(_DAdaptBase):
    """`train.optimizer: dadaptadam` (train_util.py:339-344 returns dadaptation.DAdaptAdam; requirements.txt pins
    dadaptation==3.1).  The package is absent here (parity unpinned): this follows that release's published algorithm (Defazio &
    Mishchenko 2023, "Learning-Rate-Free Learning by D-Adaptation", Adam variant, in the release-3 form whose first moment
    already carries d*lr and whose numerator is an exponential average with sqrt(beta2)), same constructor arguments and
    defaults, state tensors in the parameter dtype like the package's zeros_like(p):

        dlr    = d * lr * [sqrt(1 - beta2^(k+1)) / (1 - beta1^(k+1))  if use_bias_correction]
        acc   += dlr * <g, s / (sqrt(v) + eps)>                      (with s, v from BEFORE this step's update)
        m     <- beta1 m + dlr (1 - beta1) g ;  v <- beta2 v + (1 - beta2) g^2 ;  s <- sqrt(beta2) s + dlr (1 - sqrt(beta2)) g
        num   <- sqrt(beta2) num + (1 - sqrt(beta2)) acc
        d     <- max(d, min(num / ((1 - sqrt(beta2)) ||s||_1), d * growth_rate))
        x     <- x (1 - wd * dlr)  [decouple]  - m / (sqrt(v) + eps)

    Tensor ops on the parameter's own device, like Prodigy above (one flat buffer, a dozen launches per optimizer step)."""

    def __init__(self, params, lr: float = 1.0, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, log_every: int = 0,
                 decouple: bool = False, use_bias_correction: bool = False, d0: float = 1e-6, growth_rate: float = float("inf"),
                 fsdp_in_use: bool = False):
        if not 0.0 < d0:
            raise ValueError(f"Invalid d0 value: {d0}")
        if not 0.0 < lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 < eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if fsdp_in_use:
            raise NotImplementedError("DAdaptAdam: the adapters are replicated (data parallelism all-reduces the gradient), never "
                                      "sharded: fsdp_in_use has nothing to act on")
        if decouple:
            print("Using decoupled weight decay")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, d=d0, k=0, numerator_weighted=0.0,
                                      log_every=log_every, growth_rate=growth_rate, use_bias_correction=use_bias_correction,
                                      decouple=decouple, fsdp_in_use=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g0 = self.param_groups[0]
        beta1, beta2 = g0["betas"]
        k, d, growth_rate, decouple = g0["k"], g0["d"], g0["growth_rate"], g0["decouple"]
        lr = self._common_lr()
        bias_correction = ((1 - beta2 ** (k + 1)) ** 0.5) / (1 - beta1 ** (k + 1)) if g0["use_bias_correction"] else 1.0
        dlr = d * lr * bias_correction
        sqrt_beta2 = beta2 ** 0.5
        acc, sk_l1 = 0.0, 0.0
        for group in self.param_groups:
            decay, eps = group["weight_decay"], group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                if decay != 0 and not decouple:
                    grad.add_(p, alpha=decay)
                st = self.state[p]
                if "step" not in st:
                    st["step"] = 0
                    st["s"] = torch.zeros_like(p)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                if group["lr"] > 0.0:
                    denom = st["exp_avg_sq"].sqrt().add_(eps)
                    acc += dlr * self._dot(grad, st["s"].div(denom))
                    st["exp_avg"].mul_(beta1).add_(grad, alpha=dlr * (1 - beta1))
                    st["exp_avg_sq"].mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                    st["s"].mul_(sqrt_beta2).add_(grad, alpha=dlr * (1 - sqrt_beta2))
                    sk_l1 += self._l1(st["s"])
        numerator_weighted = sqrt_beta2 * g0["numerator_weighted"] + (1 - sqrt_beta2) * acc
        if sk_l1 == 0:              # no gradient seen: nothing to do
            return loss
        d_hat = d
        if lr > 0.0:
            d_hat = numerator_weighted / ((1 - sqrt_beta2) * sk_l1)
            d = max(d, min(d_hat, d * growth_rate))
        for group in self.param_groups:
            group["numerator_weighted"], group["d"], group["d_hat"] = numerator_weighted, d, d_hat
            decay, eps = group["weight_decay"], group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                st["step"] += 1
                denom = st["exp_avg_sq"].sqrt().add_(eps)
                if decay != 0 and decouple:
                    p.add_(p, alpha=-decay * dlr)
                p.addcdiv_(st["exp_avg"], denom, value=-1)
            group["k"] = k + 1
        return loss


class DAdaptLion(_DAdaptBase):
    """`train.optimizer: dadaptlion` (train_util.py:345-346 returns dadaptation.DAdaptLion; dadaptation==3.1, absent here:
    parity unpinned).  Lion (sign of the interpolated moment) with the D-Adaptation estimate driven by the sign update:

        dlr    = d * lr
        x     <- x (1 - dlr * wd) ;  u = sign(beta1 m + (1 - beta1) g) ;  x <- x - dlr u
        m     <- beta2 m + (1 - beta2) dlr g
        acc   += dlr * <u, s> ;  s <- sqrt(beta2) s + (1 - sqrt(beta2)) dlr u
        num   <- sqrt(beta2) num + (1 - sqrt(beta2)) acc ;  d <- max(d, num / ((1 - sqrt(beta2)) ||s||_1))

    Same constructor arguments and defaults as the package."""

    def __init__(self, params, lr: float = 1.0, betas=(0.9, 0.999), weight_decay: float = 0.0, log_every: int = 0, d0: float = 1e-6,
                 fsdp_in_use: bool = False):
        if not 0.0 < d0:
            raise ValueError(f"Invalid d0 value: {d0}")
        if not 0.0 < lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if fsdp_in_use:
            raise NotImplementedError("DAdaptLion: the adapters are replicated, never sharded: fsdp_in_use has nothing to act on")
        super().__init__(params, dict(lr=lr, betas=betas, weight_decay=weight_decay, d=d0, k=0, log_every=log_every,
                                      numerator_weighted=0.0, fsdp_in_use=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g0 = self.param_groups[0]
        beta1, beta2 = g0["betas"]
        d = g0["d"]
        lr = self._common_lr()
        dlr = d * lr
        sqrt_beta2 = beta2 ** 0.5
        acc, sk_l1 = 0.0, 0.0
        for group in self.param_groups:
            wd = group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["s"] = torch.zeros_like(p)
                if group["lr"] > 0.0:
                    p.mul_(1 - dlr * wd)
                    update = st["exp_avg"].clone().mul_(beta1).add_(grad, alpha=1 - beta1).sign_()
                    p.add_(update, alpha=-dlr)
                    st["exp_avg"].mul_(beta2).add_(grad, alpha=(1 - beta2) * dlr)
                    acc += dlr * self._dot(update, st["s"])
                    st["s"].mul_(sqrt_beta2).add_(update, alpha=(1 - sqrt_beta2) * dlr)
                    sk_l1 += self._l1(st["s"])
        numerator_weighted = sqrt_beta2 * g0["numerator_weighted"] + (1 - sqrt_beta2) * acc
        if sk_l1 == 0:
            return loss
        d_hat = d
        if lr > 0.0:
            d_hat = numerator_weighted / ((1 - sqrt_beta2) * sk_l1)
            d = max(d, d_hat)
        for group in self.param_groups:
            group["numerator_weighted"], group["d"], group["d_hat"] = numerator_weighted, d, d_hat
            group["k"] = group["k"] + 1
        return loss
