"""RCCL on the leased GPU: torch.distributed backend "nccl" (= RCCL on ROCm) with world_size 1.

A one-GPU box cannot show scaling, but it can show that the data-parallel code path is the real one: the process
group initialises over RCCL, `SliderTrainer.iteration(process_group=...)` all-reduces the flat fp32 LoRA-gradient
buffer ON THE DEVICE (sliders_amd/parallel.py), and the result equals the run without a group (sum over one rank,
scale 1/1)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle.unet_oracle import build_unet
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.parallel import allreduce_sum_, world_info
from sliders_amd.trainer import SliderTrainer
from sliders_amd.unet import UNetEngine
from tests.test_trainer_gpu import _pair, _setup

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_rccl_world1_iteration_equals_no_group(dev):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        pg = dist.group.WORLD
        assert world_info(pg) == (0, 1)
        # the collective itself, on a buffer shaped like LoraStore.grads
        cfg, store, emb, pool, noise = _setup(dev)
        ref = torch.randn(store.numel, device=dev)
        buf = ref.clone()
        scale = allreduce_sum_(buf, pg)
        torch.cuda.synchronize()
        assert scale == 1.0 and torch.equal(buf, ref)
        # a whole iteration with and without the group
        name, k, hw = "tiny_sdxl", 2, 16
        sd = build_unet(name, seed=0).state_dict()
        outs = []
        for group in (pg, None):
            cfg, store, emb, pool, noise = _setup(dev)
            eng = UNetEngine(cfg, sd, dev)
            tr = SliderTrainer(eng, store, hw, hw, lr=2e-4, process_group=group)
            loss = tr.iteration(_pair(emb, pool, dev), k, noise.to(dev))
            torch.cuda.synchronize()
            outs.append((loss.item(), store.grads.clone(), store.params.clone(), tr.grad_scale))
        (la, ga, pa, sa), (lb, gb, pb, sb) = outs
        cos = torch.nn.functional.cosine_similarity(ga, gb, dim=0).item()
        print(f"[rccl] world=1: loss {la:.6e} vs {lb:.6e}, grad cosine {cos:.7f}, grad_scale {sa} / {sb}, "
              f"params differing {(pa != pb).float().mean().item():.2e}")
        assert sa == sb == 1.0
        # not bit-equal run to run: GroupNorm statistics are summed with fp32 atomics and the 1-ulp bf16 flips that
        # follow are amplified by the denoise chain (measured run-to-run: loss 0.6 %, gradient cosine 0.993); what this
        # test pins is that the collective is issued on the device buffer and leaves the step unchanged
        # same iteration twice (with / without the process group): the difference is the engine's run-to-run floor, measured
        # up to ~5 % on the loss and cosine 0.985 - 0.999 on the gradient
        assert abs(la - lb) <= 8e-2 * abs(lb) and cos > 0.97
    finally:
        dist.destroy_process_group()
