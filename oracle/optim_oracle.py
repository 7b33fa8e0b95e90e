"""CPU oracle: the optimizers `train.optimizer` can name besides torch.optim.AdamW (which the tests call directly).

TEST INFRASTRUCTURE ONLY (see unet_oracle.py header): imported by tests/ only, never by the product path.

Lion - trainscripts/textsliders/train_util.py:365-368 returns `lion_pytorch.Lion` (requirements.txt:5 pins
lion_pytorch==0.1.2; third-party, not vendored, not installed here -> PARITY UNPINNED against the package itself).
This restates that release's `update_fn` / `Lion.step` (defaults lr 1e-4, betas (0.9, 0.99), weight_decay 0):

    p.data.mul_(1 - lr * wd)
    update = exp_avg.clone().mul_(beta1).add(grad, alpha = 1 - beta1).sign_()
    p.add_(update, alpha = -lr)
    exp_avg.mul_(beta2).add_(grad, alpha = 1 - beta2)

on tensors of the parameter dtype (bf16 in the reference's configs), so every line rounds once to bf16 exactly as
torch's tensor ops do.  Device note: torch's CPU kernel of `add(t, alpha=a)` rounds `a` to the tensor dtype, the
CUDA/ROCm kernel keeps it in fp32 opmath; the reference trains on cuda (train_lora_xl.py:414), so the bit-exact GPU
test runs this class on device tensors.  It is pinned here against a float64 evaluation of the published algorithm (Chen et al. 2023,
"Symbolic Discovery of Optimization Algorithms", Alg. 1) in tests/test_oracle.py.
"""
from __future__ import annotations

from typing import Tuple

import torch


class Lion:
    def __init__(self, params, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.99), weight_decay: float = 0.0):
        self.params = list(params)
        self.lr, self.betas, self.wd = lr, betas, weight_decay
        self.state = {}

    @torch.no_grad()
    def step(self):
        b1, b2 = self.betas
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.setdefault(id(p), {})
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p)
            exp_avg, grad = st["exp_avg"], p.grad
            p.data.mul_(1 - self.lr * self.wd)
            update = exp_avg.clone().mul_(b1).add(grad, alpha=1 - b1).sign_()
            p.add_(update, alpha=-self.lr)
            exp_avg.mul_(b2).add_(grad, alpha=1 - b2)


class ProdigyF64:
    """float64 numpy evaluation of Prodigy as released in prodigyopt 1.0 (train_util.py:369-372; package absent ->
    PARITY UNPINNED), written from the paper's Algorithm 4 plus the release's conventions, independently of
    sliders_amd/optim.py (no torch, one flat vector):

        dlr      = d * lr * [sqrt(1 - beta2^(k+1)) / (1 - beta1^(k+1))  if use_bias_correction]
        num     <- beta3 * num + (d/d0) * dlr * <g, x0 - x>                         (beta3 = sqrt(beta2))
        m       <- beta1 * m + d (1 - beta1) g ;   v <- beta2 * v + d^2 (1 - beta2) g^2
        s       <- beta3 * s + (d/d0) * (d if safeguard_warmup else dlr) * g
        d_hat    = d_coef * num / ||s||_1 ;  first growth from d0: d = max(d, d_hat) ;  d_max = max(d_max, d_hat)
        d_next   = min(d_max, d * growth_rate)
        x       <- x (1 - wd * dlr) - dlr * m / (sqrt(v) + d_next * eps)            (decoupled decay; d_next as released)
    """

    def __init__(self, x0, lr=1.0, betas=(0.9, 0.999), beta3=None, eps=1e-8, weight_decay=0.0, use_bias_correction=False,
                 safeguard_warmup=False, d0=1e-6, d_coef=1.0, growth_rate=float("inf")):
        import numpy as np
        self.np = np
        self.x = np.array(x0, dtype=np.float64)
        self.x0 = self.x.copy()
        self.m, self.v, self.s = np.zeros_like(self.x), np.zeros_like(self.x), np.zeros_like(self.x)
        self.lr, self.b1, self.b2 = lr, betas[0], betas[1]
        self.b3 = beta3 if beta3 is not None else betas[1] ** 0.5
        self.eps, self.wd, self.ubc, self.sw = eps, weight_decay, use_bias_correction, safeguard_warmup
        self.d = self.d0 = self.d_max = d0
        self.d_coef, self.growth, self.num, self.k = d_coef, growth_rate, 0.0, 0

    def step(self, g):
        np = self.np
        g = np.asarray(g, dtype=np.float64)
        d, d0 = self.d, self.d0
        bc = (1 - self.b2 ** (self.k + 1)) ** 0.5 / (1 - self.b1 ** (self.k + 1)) if self.ubc else 1.0
        dlr = d * self.lr * bc
        self.num = self.b3 * self.num + (d / d0) * dlr * float(g @ (self.x0 - self.x))
        self.m = self.b1 * self.m + d * (1 - self.b1) * g
        self.v = self.b2 * self.v + d * d * (1 - self.b2) * g * g
        self.s = self.b3 * self.s + (d / d0) * (d if self.sw else dlr) * g
        l1 = float(np.abs(self.s).sum())
        if l1 == 0:
            return
        d_hat = self.d_coef * self.num / l1
        if d == d0:
            d = max(d, d_hat)
        self.d_max = max(self.d_max, d_hat)
        d = min(self.d_max, d * self.growth)
        self.x = self.x * (1 - self.wd * dlr) - dlr * self.m / (np.sqrt(self.v) + d * self.eps)
        self.d, self.k = d, self.k + 1


class DAdaptAdamF64:
    """float64 numpy evaluation of D-Adapt Adam as released in dadaptation 3.1 (train_util.py:339-344; package absent ->
    PARITY UNPINNED), written from Defazio & Mishchenko 2023 (Alg. 5, "Adam with D-Adaptation") in the release-3
    parameterisation, independently of sliders_amd/optim.py (no torch, one flat vector).  With b = sqrt(beta2) and
    lambda_k = d_k * lr * [bias correction]:

        r_{k+1}  = b r_k + (1 - b) lambda_k <g_k, s_k / (sqrt(v_k) + eps)>          (s_k, v_k: before this step)
        m_{k+1}  = beta1 m_k + (1 - beta1) lambda_k g_k ;   v_{k+1} = beta2 v_k + (1 - beta2) g_k^2
        s_{k+1}  = b s_k + (1 - b) lambda_k g_k
        d_{k+1}  = max(d_k, min(r_{k+1} / ((1 - b) ||s_{k+1}||_1), growth d_k))
        x_{k+1}  = x_k [1 - wd lambda_k  if decoupled] - m_{k+1} / (sqrt(v_{k+1}) + eps)
    (coupled decay adds wd x_k to g_k first.)"""

    def __init__(self, x0, lr=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decouple=False, use_bias_correction=False,
                 d0=1e-6, growth_rate=float("inf")):
        import numpy as np
        self.np = np
        self.x = np.array(x0, dtype=np.float64)
        self.m, self.v, self.s = np.zeros_like(self.x), np.zeros_like(self.x), np.zeros_like(self.x)
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.decouple, self.ubc, self.d, self.growth, self.r, self.k = decouple, use_bias_correction, d0, growth_rate, 0.0, 0

    def step(self, g):
        np = self.np
        g = np.asarray(g, dtype=np.float64)
        if self.wd and not self.decouple:
            g = g + self.wd * self.x
        b = self.b2 ** 0.5
        lam = self.d * self.lr * ((1 - self.b2 ** (self.k + 1)) ** 0.5 / (1 - self.b1 ** (self.k + 1)) if self.ubc else 1.0)
        self.r = b * self.r + (1 - b) * lam * float(g @ (self.s / (np.sqrt(self.v) + self.eps)))
        self.m = self.b1 * self.m + (1 - self.b1) * lam * g
        self.v = self.b2 * self.v + (1 - self.b2) * g * g
        self.s = b * self.s + (1 - b) * lam * g
        l1 = float(np.abs(self.s).sum())
        if l1 == 0:
            return
        self.d = max(self.d, min(self.r / ((1 - b) * l1), self.d * self.growth))
        if self.wd and self.decouple:
            self.x = self.x * (1 - self.wd * lam)
        self.x = self.x - self.m / (np.sqrt(self.v) + self.eps)
        self.k += 1


class DAdaptLionF64:
    """float64 numpy evaluation of D-Adapt Lion as released in dadaptation 3.1 (train_util.py:345-346; PARITY UNPINNED): Lion's
    sign update u_k = sign(beta1 m_k + (1 - beta1) g_k) with step lambda_k = d_k lr, the moment carrying lambda_k, and the
    estimate of D-Adaptation driven by u_k instead of g_k (b = sqrt(beta2)):

        x_{k+1} = x_k (1 - lambda_k wd) - lambda_k u_k ;  m_{k+1} = beta2 m_k + (1 - beta2) lambda_k g_k
        r_{k+1} = b r_k + (1 - b) lambda_k <u_k, s_k> ;   s_{k+1} = b s_k + (1 - b) lambda_k u_k
        d_{k+1} = max(d_k, r_{k+1} / ((1 - b) ||s_{k+1}||_1))"""

    def __init__(self, x0, lr=1.0, betas=(0.9, 0.999), weight_decay=0.0, d0=1e-6):
        import numpy as np
        self.np = np
        self.x = np.array(x0, dtype=np.float64)
        self.m, self.s = np.zeros_like(self.x), np.zeros_like(self.x)
        self.lr, self.b1, self.b2, self.wd, self.d, self.r = lr, betas[0], betas[1], weight_decay, d0, 0.0

    def step(self, g):
        np = self.np
        g = np.asarray(g, dtype=np.float64)
        b, lam = self.b2 ** 0.5, self.d * self.lr
        u = np.sign(self.b1 * self.m + (1 - self.b1) * g)
        self.x = self.x * (1 - lam * self.wd) - lam * u
        self.m = self.b2 * self.m + (1 - self.b2) * lam * g
        self.r = b * self.r + (1 - b) * lam * float(u @ self.s)
        self.s = b * self.s + (1 - b) * lam * u
        l1 = float(np.abs(self.s).sum())
        if l1 == 0:
            return
        self.d = max(self.d, self.r / ((1 - b) * l1))
