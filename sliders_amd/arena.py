"""Device memory arenas.  PyTorch-ROCm owns the allocation (one big uint8 tensor per arena); the planner
bump-allocates sub-buffers from it and hands raw device pointers to the C ABI.  With 288 GB of HBM3E per
MI355X nothing is ever freed inside a plan: every intermediate of a UNet pass has its own address, so a whole
pass is a static command buffer (and hipGraph-capturable).

`device=None` gives a VIRTUAL arena (fake base address, no memory): used by the CPU tests to check planner
logic (shapes, alignment, buffer overlap) without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

_ALIGN = 256
_DT_SIZE = {torch.bfloat16: 2, torch.float32: 4, torch.int32: 4, torch.uint8: 1, torch.float16: 2, torch.int64: 8}


@dataclass
class Buf:
    ptr: int
    nbytes: int
    shape: Tuple[int, ...]
    dtype: torch.dtype
    tensor: Optional[torch.Tensor] = None  # view into the arena (real arenas only)
    name: str = ""


class Arena:
    def __init__(self, capacity: int, device=None, name: str = "arena"):
        self.capacity = int(capacity)
        self.device = device
        self.name = name
        self.off = 0
        self.high_water = 0
        self.allocs = []
        if device is not None:
            # zero-filled once: no kernel reads bytes it (or an earlier op of the plan) has not written, but a latent violation
            # of that rule must read zeros, never whatever bit patterns a previous process left in HBM
            self.buf = torch.zeros(self.capacity, dtype=torch.uint8, device=device)
            self.base = self.buf.data_ptr()
            assert self.base % _ALIGN == 0
        else:
            self.buf = None
            self.base = 0x7F0000000000  # fake, aligned

    @property
    def virtual(self) -> bool:
        return self.buf is None

    def alloc(self, shape, dtype=torch.bfloat16, name: str = "") -> Buf:
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        nbytes = n * _DT_SIZE[dtype]
        start = (self.off + _ALIGN - 1) // _ALIGN * _ALIGN
        end = start + nbytes
        if end > self.capacity:
            raise MemoryError(f"{self.name}: out of arena memory ({end} > {self.capacity}) allocating {name} {shape}")
        self.off = end
        self.high_water = max(self.high_water, end)
        t = None
        if self.buf is not None:
            t = self.buf[start:end].view(dtype).view(shape)
        b = Buf(self.base + start, nbytes, shape, dtype, t, name)
        self.allocs.append((start, end, name))
        return b

    def mark(self) -> int:
        return self.off

    def reset(self, mark: int = 0):
        self.off = mark

    def region(self, start: int, end: int):
        return self.base + start, end - start
