"""Data parallelism for slider training: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on MI355X nodes, "gloo" in the CPU tests).

The reference is single-GPU and draws ONE PromptEmbedsPair per iteration (train_lora_xl.py:170-172).  Iterations
on different pairs / noise are independent given the adapter weights, so N ranks run N of them concurrently
and exchange exactly one message per optimizer step: the flat fp32 LoRA-gradient buffer (17.3 MB for SDXL
rank 4), summed with a single all-reduce and scaled by 1/N inside the fused AdamW kernel.  Everything else is
derived locally and deterministically:
  * k (number of partial-denoise steps, 1..49) is drawn from a generator seeded identically on every rank, so
    all ranks do the same amount of work per step (otherwise the slowest rank could be 49x slower),
  * the prompt pair is a rank-dependent index into the pair list, the latent noise a rank-dependent stream,
  * adapter parameters and AdamW state are replicated and stay bit-identical because every rank applies the
    same update to the same all-reduced gradient.
N=1 reproduces the reference semantics exactly.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class StepSampler:
    """Per-iteration random choices, split into rank-shared and rank-local streams."""

    def __init__(self, seed: int, rank: int, world: int, n_pairs: int, max_denoising_steps: int = 50):
        self.rank, self.world, self.n_pairs = rank, world, n_pairs
        self.max_steps = max_denoising_steps
        self.shared = torch.Generator(device="cpu").manual_seed(seed)
        self.local = torch.Generator(device="cpu").manual_seed(seed * 1000003 + 17 + rank)
        self.step = 0

    def next(self):
        """-> (k, pair_index).  k is identical on all ranks; pair indices of one step are distinct whenever
        world <= n_pairs.  Every call consumes the SAME number of draws of the shared stream on every rank (k, the base
        pair, and the two seeds of this step's resolution / crop draws - used or not), so ranks whose prompt pairs have
        different settings (one with dynamic_resolution, one without) cannot drift apart."""
        k = int(torch.randint(1, self.max_steps, (1,), generator=self.shared).item())
        base = int(torch.randint(0, self.n_pairs, (1,), generator=self.shared).item())
        self.res_seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self.shared).item())
        self.crop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self.shared).item())
        pair = (base + self.rank) % self.n_pairs
        self.step += 1
        return k, pair

    def resolution(self, settings):
        """(height, width) of this step for a prompt pair's settings (train_lora_xl.py:179-186): the bucket draw of
        get_random_resolution_in_bucket under this step's rank-shared seed, so every rank whose pair asks for the same
        bucket denoises the same latent size (equal work per step)."""
        from .train_util import get_random_resolution_in_bucket
        height = width = settings.resolution
        if settings.dynamic_resolution:
            st = torch.random.get_rng_state()
            torch.manual_seed(self.res_seed)
            height, width = get_random_resolution_in_bucket(settings.resolution)
            torch.random.set_rng_state(st)
        return height, width

    def time_ids(self, settings, height, width, is_xl):
        """SDXL micro-conditioning with dynamic_crops (train_lora_xl.py:188-203) under this step's rank-shared seed, or None."""
        if not (is_xl and settings.dynamic_crops):
            return None
        from .train_util import get_add_time_ids
        st = torch.random.get_rng_state()
        torch.manual_seed(self.crop_seed)
        ids = get_add_time_ids(height, width, dynamic_crops=True, dtype=torch.bfloat16)   # bf16 like the reference (quirk D.8)
        torch.random.set_rng_state(st)
        return ids.float().repeat(2 * settings.batch_size, 1)

    def noise(self, shape) -> torch.Tensor:
        """Latent noise, drawn on the CPU like the reference (train_util.py:20-32)."""
        return torch.randn(shape, generator=self.local)


def allreduce_sum_(flat: torch.Tensor, group=None) -> float:
    """In-place SUM all-reduce of the flat gradient buffer; returns the factor the optimizer must scale by.
    With an explicit `group` the collective is issued even for a single rank (an identity there), so the RCCL path
    can be exercised on a one-GPU box."""
    rank, world = world_info(group)
    if world > 1 or (group is not None and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def broadcast_params_(flat: torch.Tensor, group=None, src: int = 0):
    """Make rank-local adapter init identical (only needed if ranks seeded their init differently)."""
    rank, world = world_info(group)
    if world > 1:
        dist.broadcast(flat, src=src, group=group)
