"""Data-parallel path, world_size 2, gloo backend on CPU (the production backend is "nccl" = RCCL over xGMI).

Checks the three properties the DP design relies on (sliders_amd/parallel.py):
  * k (denoise length) is identical on all ranks, prompt pairs differ, noise streams differ;
  * one all-reduce of the flat fp32 gradient buffer + the 1/N scale equals the gradient of the summed
    per-rank losses divided by N (here: a small quadratic "adapter" whose gradient is known in closed form);
  * replicated parameters stay bit-identical across ranks after the update.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sliders_amd.parallel import StepSampler, allreduce_sum_, broadcast_params_, world_info


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert world_info() == (rank, world)
        samp = StepSampler(seed=7, rank=rank, world=world, n_pairs=8)
        ks, pairs = zip(*[samp.next() for _ in range(16)])
        noise = samp.noise((1, 4, 8, 8))
        # replicated parameters, per-rank data
        torch.manual_seed(0)
        w = torch.randn(1000)
        broadcast_params_(w)
        data = torch.randn(1000, generator=torch.Generator().manual_seed(100 + rank))
        grad = (w - data).float()                 # d/dw 0.5*|w - data|^2 on this rank
        scale = allreduce_sum_(grad)
        w_new = w - 0.1 * grad * scale
        # the real exchange: LoraStore's flat fp32 gradient buffer (what SliderTrainer.iteration all-reduces), then the
        # replicated AdamW update with the 1/N scale folded in (torch.optim.AdamW stands in for slh_adamw on the CPU;
        # tests/test_kernels_gpu.py::test_adamw_bit_exact ties the two together)
        from sliders_amd.config import CONFIGS
        from sliders_amd.lora_store import LoraStore
        torch.manual_seed(3)
        store = LoraStore(CONFIGS["tiny_sdxl"](), rank=4, alpha=1.0, train_method="noxattn", device="cpu")
        store.grads.copy_(torch.randn(store.numel, generator=torch.Generator().manual_seed(200 + rank)))
        gscale = allreduce_sum_(store.grads)
        prm = torch.nn.Parameter(store.params.float().clone())
        opt = torch.optim.AdamW([prm], lr=2e-4)
        prm.grad = store.grads * gscale
        opt.step()
        q.put((rank, list(ks), list(pairs), noise.sum().item(), grad.numpy().copy(), scale, w_new.numpy().copy(),
               store.grads.numpy().copy(), prm.detach().numpy().copy(), store.numel))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_world(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=100) for _ in range(world)], key=lambda r: r[0])
        for p in procs:
            p.join(30)
            assert p.exitcode == 0
        return res
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()


@pytest.mark.timeout(300)
def test_dp_world2_gloo():
    world = 2
    try:
        res = _run_world(world)
    except Exception:           # the probed rendezvous port can be taken between probe and bind: one retry
        res = _run_world(world)
    (r0, k0, p0, n0, g0, s0, w0, lg0, lp0, numel), (r1, k1, p1, n1, g1, s1, w1, lg1, lp1, _) = res
    # payloads travel as numpy arrays (pickled by value; tensors travel by fd and need the sender alive)
    g0, g1, w0, w1, lg0, lg1, lp0, lp1 = (torch.from_numpy(a) for a in (g0, g1, w0, w1, lg0, lg1, lp0, lp1))
    assert k0 == k1, "denoise length must be shared across ranks"
    assert all(1 <= k <= 49 for k in k0)
    assert all(a != b for a, b in zip(p0, p1)), "ranks must train different prompt pairs in a step"
    assert n0 != n1, "ranks must draw different latent noise"
    assert s0 == s1 == 0.5
    assert torch.equal(g0, g1), "all-reduced gradient must be identical on every rank"
    torch.manual_seed(0)
    w = torch.randn(1000)
    d0 = torch.randn(1000, generator=torch.Generator().manual_seed(100))
    d1 = torch.randn(1000, generator=torch.Generator().manual_seed(101))
    assert torch.allclose(g0, (w - d0) + (w - d1))
    assert torch.equal(w0, w1), "replicated parameters diverged"
    # LoraStore-shaped buffer: sum of the two ranks' gradients on both ranks, identical AdamW result
    e0 = torch.randn(numel, generator=torch.Generator().manual_seed(200))
    e1 = torch.randn(numel, generator=torch.Generator().manual_seed(201))
    assert numel > 100000 and torch.equal(lg0, lg1) and torch.allclose(lg0, e0 + e1)
    assert torch.equal(lp0, lp1), "replicated adapter parameters diverged after the optimizer step"


def test_single_process_is_identity():
    g = torch.arange(10, dtype=torch.float32)
    assert allreduce_sum_(g) == 1.0 and torch.equal(g, torch.arange(10, dtype=torch.float32))
    s = StepSampler(seed=1, rank=0, world=1, n_pairs=2)
    k, p = s.next()
    assert 1 <= k <= 49 and p in (0, 1)


def test_step_sampler_ranks_stay_in_step_with_mixed_prompt_settings():
    """Ranks whose prompt pairs have DIFFERENT settings (dynamic_resolution / dynamic_crops on one pair, off on another) must keep
    drawing the same k - and the same (height, width) whenever their pairs ask for the same bucket - at every step: the shared
    stream is consumed identically by every rank whatever its pair needs (sliders_amd/parallel.py StepSampler.next).  A rank
    that consumed one draw more would run a different denoise length from the next step on (up to 49x the work)."""
    from types import SimpleNamespace as S
    settings = [S(resolution=1024, dynamic_resolution=True, dynamic_crops=True, batch_size=1),
                S(resolution=1024, dynamic_resolution=False, dynamic_crops=False, batch_size=1),
                S(resolution=1024, dynamic_resolution=True, dynamic_crops=False, batch_size=1),
                S(resolution=512, dynamic_resolution=True, dynamic_crops=True, batch_size=2)]
    world = 4
    samps = [StepSampler(seed=11, rank=r, world=world, n_pairs=len(settings)) for r in range(world)]
    seen_dynamic = set()
    for step in range(64):
        draws = []
        for smp in samps:
            k, pi = smp.next()
            s = settings[pi]
            h, w = smp.resolution(s)
            ids = smp.time_ids(s, h, w, is_xl=True)
            draws.append((k, pi, h, w, None if ids is None else tuple(ids.flatten().tolist()), s))
        assert len({d[0] for d in draws}) == 1, f"step {step}: k differs across ranks: {[d[0] for d in draws]}"
        assert len({d[1] for d in draws}) == world, "pair indices of one step are distinct while world <= n_pairs"
        # the ranks whose pairs draw from the same bucket get the same resolution (equal work), fixed-resolution pairs their own
        dyn1024 = {(d[2], d[3]) for d in draws if d[5].dynamic_resolution and d[5].resolution == 1024}
        assert len(dyn1024) == 1
        seen_dynamic |= dyn1024
        for d in draws:
            if not d[5].dynamic_resolution:
                assert (d[2], d[3]) == (d[5].resolution, d[5].resolution)
            assert (d[4] is not None) == d[5].dynamic_crops
            assert 0 < d[2] <= d[5].resolution and d[2] % 64 == 0
    assert len(seen_dynamic) > 4, "the bucket draw varies from step to step"
    # the global torch RNG (the reference's own stream, e.g. for dropout-free init) is left untouched by the shared draws
    torch.manual_seed(123)
    a = torch.rand(3)
    torch.manual_seed(123)
    samps[0].next(); samps[0].resolution(settings[0]); samps[0].time_ids(settings[0], 512, 512, True)
    assert torch.equal(a, torch.rand(3))
