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
        # every reduction of the iteration runs in a fixed order (forward: round 3; the M-split sums of the weight gradients:
        # round 4) and a one-rank all-reduce is the identity: loss, gradient buffer and updated parameters are EQUAL
        assert la == lb
        assert torch.equal(ga, gb), f"{int((ga != gb).sum())} gradient elements differ with / without the process group"
        assert torch.equal(pa, pb)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_bench_under_torchrun_one_rank(dev):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), on the
    one GPU this box has: process group by LOCAL_RANK / device_id over RCCL, barrier-bracketed timing, max over ranks, one
    JSON line from rank 0, clean teardown.  BENCH_FORCE_PROCESS_GROUP makes world_size 1 take the N > 1 code path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_FORCE_PROCESS_GROUP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--model", "sd1", "--res", "256", "--no-cpu-baseline", "--no-roofline", "--no-extra"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["value"] > 0 and res["config"]["parallelism"] == "dp1"
    assert res["unit"] == "steps/s" and res["scaling"] == "weak" and res["higher_is_better"] is True
    # the contract the driver's SCALE run relies on: n_gpus is WORLD_SIZE, the step count is the sum over ranks (every rank runs the
    # same k per iteration: the rank-shared stream), value = that sum / the max-over-ranks wall time
    cfg_ = res["config"]
    world = int(res["n_gpus"])
    assert cfg_["unet_denoise_steps_timed"] % world == 0 and cfg_["unet_denoise_steps_timed"] >= world * res["steps"] * 5
    assert abs(res["value"] - cfg_["unet_denoise_steps_timed"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 0.02 * res["value"]
    assert abs(cfg_["iterations_per_s"] - world * 1e3 / res["ms_per_step"]) < 0.02 * cfg_["iterations_per_s"]
    # round 5: the run checks itself - every rank's adapter replica digest is all-gathered after the timed region and must be equal,
    # and every gradient all-reduce of the timed loop is bracketed by events on the stream it is ordered on (compute / comm split)
    dp = res["data_parallel"]
    assert dp["rccl_ranks_seen"] == world and dp["replicas_bit_identical"] is True and dp["adapter_state_digest"].startswith("0x")
    assert dp["allreduces_timed_per_rank"] == res["steps"] and len(dp["allreduce_us_mean_per_rank"]) == world
    assert 0 < dp["allreduce_us_mean_per_rank"][0] < 5e4 and dp["allreduce_bytes"] > 0
