"""Parity hardening (round 4): fewer links between the reference's code and the HIP path, and iteration-level bounds that are
EARNED by a reference-precision arm instead of asserted.

  1. test_engine_matches_reference_fixture_directly: the HIP engine driven through the drop-in objects
     (sliders_amd.lora.LoRANetwork, `with network`, sliders_amd.train_util.predict_noise[_xl] / diffusion[_xl]) against
     tests/golden/tiny_forward.pt - the outputs of the reference's OWN lora.py / train_util.py (fp32) stored by
     tests/golden/make_golden.py.  No oracle forward runs in this test: oracle.unet_oracle.build_unet only re-creates the seeded
     weights the fixture was generated with (the engine loads that state_dict).
  2. test_iteration_against_fp32_and_bf16_arm: one whole training iteration (train_lora_xl.py:162-356) three ways - the fused HIP
     trainer, the reference loop on the fp32 oracle, the same loop on the oracle in torch bf16 (what the reference itself runs:
     config.yaml precision bfloat16).  The engine must be no further from fp32 than the bf16 arm is (x1.5 + a floor): the
     tolerance is measured beside the product, on the same inputs.
  3. test_training_dynamics_20_steps: 20 optimizer steps (k cycling 1..3, fresh seeded noise per step) of the fused trainer vs
     the reference loop + torch.optim.AdamW on the fp32 oracle: loss trajectories agree step by step, both decrease the same
     windowed loss, the accumulated LoRA update points the same way.  The offline stand-in for "trained sliders reproduce the
     reference's CLIP-score direction" (no checkpoints / CLIP weights exist here).
  4. test_full_width_iteration_sdxl_512: item 2's fp32 comparison at the REAL SDXL width (1280 channels, 20 heads, 10-layer
     transformers) on a 64 x 64 latent, k = 1 - one denoise pass, the B = 3 frozen pass, training forward, backward, AdamW.
"""
import os
import time

import pytest
import torch
import torch.nn.functional as F

from oracle.ddim_oracle import DDIMScheduler
from oracle.lora_oracle import LoRANetworkOracle
from oracle.unet_oracle import build_unet
from sliders_amd import lora as sl_lora
from sliders_amd import train_util
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.trainer import PairEmbeds, SliderTrainer
from sliders_amd.unet import UNetEngine
from tests.util import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ----------------------------------------------------------------------------------------------------------------------------------
# 1. HIP vs the reference-generated fixture, directly
# ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key", ["tiny_sdxl/noxattn", "tiny_sd1/noxattn", "tiny_sdxl/full"])
def test_engine_matches_reference_fixture_directly(dev, key):
    e = torch.load(os.path.join(G, "tiny_forward.pt"))[key]
    name, method = key.split("/")
    cfg = CONFIGS[name]()
    if "ResnetBlock2D" not in sl_lora.DEFAULT_TARGET_REPLACE:       # c3lier, as make_golden.py switched it on (train_lora_xl.py:50-52)
        sl_lora.DEFAULT_TARGET_REPLACE += sl_lora.UNET_TARGET_REPLACE_MODULE_CONV
    unet = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)      # the seeded weights of the fixture; no oracle forward
    unet.requires_grad_(False)
    network = sl_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method=method).to(dev, dtype=torch.bfloat16)
    network.store.load_state_dict(e["lora_state_dict"])
    to = lambda x: x.to(dev, torch.bfloat16)
    lat, ctx, t = to(e["latents"]), to(e["ctx"]), e["t"]
    xl = "pooled" in e
    sched = train_util.DDIMScheduler()
    sched.set_timesteps(50, device=dev)
    tt = torch.tensor(t)

    def raw(x2):
        if xl:
            return unet(x2, tt, encoder_hidden_states=ctx,
                        added_cond_kwargs={"text_embeds": to(e["pooled"]), "time_ids": to(e["time_ids"])}).sample
        return unet(x2, tt, encoder_hidden_states=ctx).sample

    with torch.no_grad():
        with network:
            eps_on = raw(torch.cat([lat] * 2))
            if xl:
                pred = train_util.predict_noise_xl(unet, sched, tt, lat, text_embeddings=ctx, add_text_embeddings=to(e["pooled"]),
                                                   add_time_ids=to(e["time_ids"]), guidance_scale=3)
                den = train_util.diffusion_xl(unet, sched, lat, text_embeddings=ctx, add_text_embeddings=to(e["pooled"]),
                                              add_time_ids=to(e["time_ids"]), guidance_scale=3, total_timesteps=3, start_timesteps=0)
            else:
                pred = train_util.predict_noise(unet, sched, tt, lat, ctx, guidance_scale=3)
                den = train_util.diffusion(unet, sched, lat, ctx, total_timesteps=3, start_timesteps=0, guidance_scale=3)
        eps_off = raw(torch.cat([lat] * 2))
    torch.cuda.synchronize()
    out = {}
    for nm, got, ref in (("eps_on", eps_on, e["eps_on"]), ("eps_off", eps_off, e["eps_off"]), ("pred_on_g3", pred, e["pred_on_g3"]),
                         ("denoised_3", den, e["denoised_3"])):
        got = got.float().cpu()
        r, m = rel_err(got, ref), (got - ref).abs().max().item()
        out[nm] = (r, m)
        print(f"[parity] fixture {key} {nm}: HIP (bf16) vs reference-generated fp32 fixture rel_l2={r:.3e} max_abs={m:.3e} "
              f"ref_rms={ref.pow(2).mean().sqrt():.3f}")
        assert torch.isfinite(got).all()
    # the adapters matter in the fixture (|eps_on - eps_off| is several bf16 ulps), so a switched-off or mis-ordered adapter fails here
    eff = rel_err(e["eps_on"], e["eps_off"])
    eff_hip = rel_err(eps_on.float().cpu(), eps_off.float().cpu())
    print(f"[parity] fixture {key}: adapter effect rel_l2 reference {eff:.3e} / HIP {eff_hip:.3e}")
    assert abs(eff_hip - eff) < 0.25 * eff + 2e-3
    # bf16 storage of a unit-variance epsilon resolves 2^-9 relative per element (rel-L2 floor ~2.3e-3 for ONE rounding); the tiny
    # nets' bf16 chains measure 0.8-1.6e-2 against fp32 for the torch-bf16 arm as well (tests/test_unet_gpu.py).  Bounds = 1.3 x the
    # values measured on MI355X (profiles/r05_parity_lines.txt: every [parity] line of the GPU suite); guidance 3 amplifies the CFG difference by 3.
    assert out["eps_on"][0] < 2.0e-2 and out["eps_off"][0] < 2.0e-2
    assert out["pred_on_g3"][0] < 4.6e-2
    assert out["denoised_3"][0] < 1.5e-2


# ----------------------------------------------------------------------------------------------------------------------------------
# shared: the reference loop on the oracle, any dtype
# ----------------------------------------------------------------------------------------------------------------------------------
def _setup(name, seed=5, hw=16, up_std=0.03):
    cfg = CONFIGS[name]()
    g = torch.Generator().manual_seed(seed)
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in ("target", "positive", "neutral", "uncond")}
    pool = {k: (torch.randn(1, cfg.pooled_dim, generator=g) if cfg.is_xl else None) for k in emb}
    noise = torch.randn(1, 4, hw, hw, generator=g)
    return cfg, emb, pool, noise, g


def _pair(emb, pool, dev, action, gs=4.0):
    cat = lambda x: torch.cat([emb["uncond"], x]).to(dev, torch.bfloat16).contiguous()
    pc = lambda x: None if x is None else torch.cat([pool["uncond"], x]).to(dev, torch.bfloat16).contiguous()
    return PairEmbeds(cat(emb["target"]), cat(emb["positive"]), cat(emb["neutral"]), cat(emb["uncond"]),
                      pc(pool["target"]), pc(pool["positive"]), pc(pool["neutral"]), pc(pool["uncond"]),
                      guidance_scale=gs, action=action)


def _ref_iteration(net, nw, cfg, emb, pool, noise, k, action, gs, dtype, hw, device="cpu"):
    """train_lora_xl.py:162-345 on the oracle in `dtype`; inputs are bf16-rounded first (every arm sees the same numbers).
    Leaves the gradients in nw's parameters; returns (denoised, target eps, loss) as fp32 CPU tensors."""
    r = lambda a: a.to(torch.bfloat16).to(device=device, dtype=dtype)
    sch = DDIMScheduler()
    tid = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * 2)

    def predict(x, which, t, g):
        ctx = r(torch.cat([emb["uncond"], emb[which]]))
        kw = {"text_embeds": r(torch.cat([pool["uncond"], pool[which]])), "time_ids": r(tid)} if cfg.is_xl else None
        e = net(torch.cat([x] * 2), t.to(device), ctx, kw).sample
        u, c = e.chunk(2)
        return u + g * (c - u)

    with torch.no_grad():
        sch.set_timesteps(50)
        x = r(noise)
        with nw:
            for t in sch.timesteps[0:k]:
                x = sch.step(predict(x, "target", t, 3), t, x).prev_sample.to(dtype)
        sch.set_timesteps(1000)
        t_cur = sch.timesteps[int(k * 1000 / 50)]
        pos, neu, unc = (predict(x, w, t_cur, 1) for w in ("positive", "neutral", "uncond"))
    with nw:
        tgt = predict(x, "target", t_cur, 1)
    sign = 1.0 if action == "enhance" else -1.0
    loss = F.mse_loss(tgt, neu + sign * gs * (pos - unc))
    loss.backward()
    return x.float().cpu(), tgt.detach().float().cpu(), loss.detach().float().cpu()


def _flat_grads(store, nw):
    flat = torch.zeros(store.numel)
    mods = {m.lora_name: m for m in nw.unet_loras}
    for e in store.entries:
        m = mods[e.name]
        flat[e.down_off:e.down_off + e.down_numel] = store._down_to_kernel(e, m.lora_down.weight.grad.float().cpu())
        flat[e.up_off:e.up_off + e.up_numel] = m.lora_up.weight.grad.float().cpu().reshape(-1)
    return flat


def _oracle_with_lora(name, sd, dtype, device="cpu"):
    net = build_unet(name, seed=0).to(device=device, dtype=dtype)
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    nw.load_state_dict(sd, strict=True)
    nw.to(device=device, dtype=dtype)
    for p_ in nw.parameters():
        p_.requires_grad_(True)
    return net, nw


# ----------------------------------------------------------------------------------------------------------------------------------
# 2. one iteration: engine vs fp32 oracle vs torch-bf16 arm
# ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,action", [("tiny_sdxl", "enhance"), ("tiny_sd1", "erase")])
def test_iteration_against_fp32_and_bf16_arm(dev, name, action):
    k, hw, gs = 3, 16, 4.0
    seeds = (5, 6, 7, 8)
    sq = {"engine": [0.0] * 5, "bf16 arm": [0.0] * 5}
    for seed in seeds:
        cfg, emb, pool, noise, g = _setup(name, seed=seed)
        store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
        for e in store.entries:
            store.params[e.up_off:e.up_off + e.up_numel] = (torch.randn(e.up_numel, generator=g) * 0.03).to(dev, torch.bfloat16)
        sd = store.state_dict()
        eng = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)
        tr = SliderTrainer(eng, store, hw, hw, lr=2e-4)
        loss_e = tr.iteration(_pair(emb, pool, dev, action, gs), k, noise.to(dev)).item()
        torch.cuda.synchronize()
        den_e, tgt_e, g_e = tr.denoised.float().cpu(), tr.e_tgt.float().cpu(), store.grads.float().cpu()
        arms = {}
        for nm, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            net, nw = _oracle_with_lora(name, sd, dtype)
            den, tgt, loss = _ref_iteration(net, nw, cfg, emb, pool, noise, k, action, gs, dtype, hw)
            arms[nm] = (den, tgt, loss.item(), _flat_grads(store, nw))
        d32, t32, l32, g32 = arms["fp32"]
        dbf, tbf, lbf, gbf = arms["bf16"]
        for nm, (den, tgt, loss, gr) in (("engine", (den_e, tgt_e, loss_e, g_e)), ("bf16 arm", (dbf, tbf, lbf, gbf))):
            r = (rel_err(den, d32), rel_err(tgt, t32), abs(loss - l32) / l32, 1.0 - F.cosine_similarity(gr, g32, dim=0).item(),
                 rel_err(gr, g32))
            print(f"[parity] iteration {name} seed {seed} {nm:8s} vs fp32 loop: denoised rel_l2={r[0]:.3e} target-eps rel_l2={r[1]:.3e} "
                  f"loss rel={r[2]:.3e} grad 1-cos={r[3]:.3e} grad rel_l2={r[4]:.3e}")
            for i in range(5):
                sq[nm][i] += r[i] * r[i] / len(seeds)
    e = [v ** 0.5 for v in sq["engine"]]
    b = [v ** 0.5 for v in sq["bf16 arm"]]
    print(f"[parity] iteration {name} RMS over {len(seeds)} seeds, engine / bf16 arm: denoised {e[0]:.3e} / {b[0]:.3e}, target eps "
          f"{e[1]:.3e} / {b[1]:.3e}, loss {e[2]:.3e} / {b[2]:.3e}, grad 1-cos {e[3]:.3e} / {b[3]:.3e}, grad rel_l2 {e[4]:.3e} / {b[4]:.3e}")
    # the engine is at most 1.5 x the reference-precision arm away from exact arithmetic on every quantity of the iteration, in the
    # RMS over the seeds (the loss is a difference of four predictions that each carry ~1e-2 of bf16 noise: a single sample of its
    # error scatters by a factor of several in BOTH arms - one draw must not decide a test)
    assert e[0] <= 1.5 * b[0] + 1e-3, "denoised latents"
    assert e[1] <= 1.5 * b[1] + 1e-3, "target prediction"
    assert e[2] <= 1.5 * b[2] + 5e-3, "loss"
    assert e[3] <= 1.5 * b[3] + 2e-3, "gradient direction"
    assert e[4] <= 1.5 * b[4] + 1e-2, "gradient"


# ----------------------------------------------------------------------------------------------------------------------------------
# 3. training dynamics over 20 optimizer steps
# ----------------------------------------------------------------------------------------------------------------------------------
def test_training_dynamics_20_steps(dev):
    name, action, hw, gs, steps, lr = "tiny_sdxl", "enhance", 16, 4.0, 20, 2e-3
    cfg, emb, pool, _, g = _setup(name, seed=21)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)     # the reference's init: lora_up = 0
    params0 = store.params.clone()
    sd0 = store.state_dict()
    noises = [torch.randn(1, 4, hw, hw, generator=g) for _ in range(steps)]
    ks = [1 + (i % 3) for i in range(steps)]
    eng = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)
    tr = SliderTrainer(eng, store, hw, hw, lr=lr)
    pair = _pair(emb, pool, dev, action, gs)
    loss_e = [tr.iteration(pair, ks[i], noises[i].to(dev)).item() for i in range(steps)]
    torch.cuda.synchronize()
    # the reference loop + the reference's optimizer (torch.optim.AdamW, train_util.py:336-373 default) on the fp32 oracle
    net, nw = _oracle_with_lora(name, sd0, torch.float32)
    opt = torch.optim.AdamW(nw.prepare_optimizer_params() if hasattr(nw, "prepare_optimizer_params") else nw.parameters(), lr=lr)
    loss_r = []
    for i in range(steps):
        opt.zero_grad()
        _, _, l = _ref_iteration(net, nw, cfg, emb, pool, noises[i], ks[i], action, gs, torch.float32, hw)
        opt.step()
        loss_r.append(l.item())
    # parameters of both runs in the kernel's flat layout
    flat_r = torch.zeros(store.numel)
    mods = {m.lora_name: m for m in nw.unet_loras}
    for e in store.entries:
        m = mods[e.name]
        flat_r[e.down_off:e.down_off + e.down_numel] = store._down_to_kernel(e, m.lora_down.weight.detach())
        flat_r[e.up_off:e.up_off + e.up_numel] = m.lora_up.weight.detach().reshape(-1)
    d_e = (store.params.float() - params0.float()).cpu()
    d_r = flat_r - params0.float().cpu()
    cos = F.cosine_similarity(d_e, d_r, dim=0).item()
    up = torch.zeros(store.numel, dtype=torch.bool)
    for e in store.entries:
        up[e.up_off:e.up_off + e.up_numel] = True
    cos_up = F.cosine_similarity(d_e[up], d_r[up], dim=0).item()
    traj = max(abs(a - b) / b for a, b in zip(loss_e, loss_r))
    first_e, last_e = sum(loss_e[:5]) / 5, sum(loss_e[-5:]) / 5
    first_r, last_r = sum(loss_r[:5]) / 5, sum(loss_r[-5:]) / 5
    print(f"[parity] dynamics: loss engine {loss_e[0]:.4e} -> {loss_e[-1]:.4e} (5-step means {first_e:.4e} -> {last_e:.4e}), oracle "
          f"{loss_r[0]:.4e} -> {loss_r[-1]:.4e} ({first_r:.4e} -> {last_r:.4e}); max step-wise rel diff {traj:.3e}; "
          f"LoRA delta cosine {cos:.4f} (up matrices {cos_up:.4f}), |delta| {d_e.norm():.3e} vs {d_r.norm():.3e}")
    assert all(l == l and l < 1e3 for l in loss_e)
    assert traj < 0.10, "the two loss trajectories stay within 10 % of each other at every step"
    assert (last_e < first_e) == (last_r < first_r), "both runs move the windowed loss the same way"
    assert abs((last_e / first_e) - (last_r / first_r)) < 0.08
    assert cos > 0.90 and cos_up > 0.90, "the accumulated update points the way the reference's does"
    assert 0.8 < d_e.norm().item() / d_r.norm().item() < 1.25


# ----------------------------------------------------------------------------------------------------------------------------------
# 4. one iteration at full SDXL width
# ----------------------------------------------------------------------------------------------------------------------------------
def test_full_width_iteration_sdxl_512(dev):
    from sliders_amd.random_init import random_state_dict
    name, action, hw, gs, k = "sdxl", "enhance", 64, 4.0, 1
    cfg = CONFIGS[name]()
    t0 = time.time()
    sd_unet = random_state_dict(cfg, dev, 0, torch.bfloat16)        # every arm multiplies with the same bf16-rounded weights
    g = torch.Generator().manual_seed(31)
    emb = {n: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for n in ("target", "positive", "neutral", "uncond")}
    pool = {n: torch.randn(1, cfg.pooled_dim, generator=g) for n in emb}
    noise = torch.randn(1, 4, hw, hw, generator=g)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    like = (torch.randn(store.numel, generator=g) * 0.05).to(torch.bfloat16)
    for e in store.entries:
        store.params[e.up_off:e.up_off + e.up_numel] = like[e.up_off:e.up_off + e.up_numel].to(dev)
    sd = store.state_dict()
    eng = UNetEngine(cfg, sd_unet, dev)
    tr = SliderTrainer(eng, store, hw, hw, lr=2e-4)
    loss_e = tr.iteration(_pair(emb, pool, dev, action, gs), k, noise.to(dev)).item()
    torch.cuda.synchronize()
    den_e, tgt_e, g_e = tr.denoised.float().cpu(), tr.e_tgt.float().cpu(), store.grads.float().cpu()
    del tr, eng
    torch.cuda.empty_cache()
    # fp32 reference loop with the oracle on the GPU through torch ops, MIOpen off (the checker only; 5 forwards + 1 backward at
    # full width take minutes on the boxes' 16 host cores)
    net = build_unet(name, device="meta")
    net.load_state_dict({k_: v.to(device=dev, dtype=torch.float32) for k_, v in sd_unet.items()}, assign=True)
    net.requires_grad_(False)
    net.eval()
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    nw.load_state_dict(sd, strict=True)
    nw.to(device=dev, dtype=torch.float32)
    for p_ in nw.parameters():
        p_.requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        den, tgt, loss = _ref_iteration(net, nw, cfg, emb, pool, noise, k, action, gs, torch.float32, hw, device=dev)
    g32 = _flat_grads(store, nw)
    r_den, r_tgt = rel_err(den_e, den), rel_err(tgt_e, tgt)
    cos = F.cosine_similarity(g_e, g32, dim=0).item()
    print(f"[parity] full-width SDXL 512^2 iteration (k={k}): denoised rel_l2={r_den:.3e} target-eps rel_l2={r_tgt:.3e} "
          f"loss {loss_e:.5e} vs fp32 {loss.item():.5e} ({abs(loss_e - loss.item()) / loss.item():.2e}) grad cosine {cos:.5f} "
          f"grad rel_l2={rel_err(g_e, g32):.3e}  [{time.time() - t0:.0f} s]")
    # measured on MI355X: 3.6e-3 / 9.0e-3 / 0.56 % / cosine 1.0000 / gradient rel-L2 1.6e-2 (profiles/r05_parity_lines.txt: every [parity] line of the GPU suite); bounds 1.3 x
    assert r_den < 5e-3 and r_tgt < 1.2e-2
    assert abs(loss_e - loss.item()) < 0.015 * loss.item()
    assert cos > 0.998 and rel_err(g_e, g32) < 2.5e-2
