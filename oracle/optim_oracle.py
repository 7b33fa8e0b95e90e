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
