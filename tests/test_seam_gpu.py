"""The seam the reference actually binds, run literally:

    unet = UNetEngine(...); network = LoRANetwork(unet, ...); optimizer = torch.optim.AdamW(network.prepare_optimizer_params())
    with network: denoised = train_util.diffusion_xl(...)            # train_lora_xl.py:205-227
    positive / neutral / unconditional = train_util.predict_noise_xl(...)   # adapters off, :236-295
    with network: target = train_util.predict_noise_xl(...)          # grad enabled, :302-322
    loss = prompt_pair.loss(...); loss.backward(); optimizer.step()  # :331-346
    network.save_weights(path)                                       # -> strict load into the reference-shaped LoRANetwork

i.e. the object-level drop-in: `sliders_amd.lora.LoRANetwork`, `with network:`, `sliders_amd.train_util`,
`sliders_amd.prompt_util.PromptEmbedsPair.loss`, `loss.backward()` through the engine's autograd bridge and a stock
torch optimizer on the flat parameter.  It must agree with the fused `SliderTrainer.iteration` on the same inputs and
its checkpoint must strict-load into the oracle restatement of the reference's LoRANetwork (lora.py:103-112, 249-258).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle.lora_oracle import LoRANetworkOracle
from oracle.unet_oracle import build_unet
from sliders_amd import lora as sl_lora
from sliders_amd import prompt_util, train_util
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.trainer import PairEmbeds, SliderTrainer
from tests.util import rel_err
from sliders_amd.unet import UNetEngine
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,action", [("tiny_sdxl", "enhance"), ("tiny_sd1", "erase")])
def test_reference_shaped_loop_equals_fused_iteration(dev, tmp_path, name, action):
    cfg = CONFIGS[name]()
    hw, k, gs = 16, 3, 4.0
    g = torch.Generator().manual_seed(5)
    emb = {n: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for n in ("target", "positive", "neutral", "uncond")}
    pool = {n: (torch.randn(1, cfg.pooled_dim, generator=g) if cfg.is_xl else None) for n in emb}
    noise = torch.randn(1, 4, hw, hw, generator=g)
    net = build_unet(name, seed=0)

    # the reference's train scripts switch c3lier on by extending this list in place (train_lora_xl.py:50-52)
    if "ResnetBlock2D" not in sl_lora.DEFAULT_TARGET_REPLACE:
        sl_lora.DEFAULT_TARGET_REPLACE += sl_lora.UNET_TARGET_REPLACE_MODULE_CONV

    # ---------------- A: the reference-shaped loop over the drop-in objects ----------------
    unet = UNetEngine(cfg, net.state_dict(), dev)
    unet.requires_grad_(False)
    unet.eval()
    torch.manual_seed(9)
    network = sl_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn").to(dev, dtype=torch.bfloat16)
    up_g = torch.Generator().manual_seed(3)
    st = network.store
    for e in st.entries:   # non-zero up weights so the partial denoise actually depends on the adapters
        st.params[e.up_off:e.up_off + e.up_numel] = (torch.randn(e.up_numel, generator=up_g) * 0.03).to(dev, torch.bfloat16)
    params0 = st.params.clone()
    optimizer = torch.optim.AdamW(network.prepare_optimizer_params(), lr=2e-4)
    criteria = torch.nn.MSELoss()
    sched = train_util.DDIMScheduler()
    settings = prompt_util.PromptSettings(target="t", positive="p", neutral="n", unconditional="", action=action,
                                          guidance_scale=gs, resolution=hw * 8, batch_size=1)
    to = lambda x: x.to(dev, torch.bfloat16)
    pair = prompt_util.PromptEmbedsPair(criteria, *(to(emb[n]) for n in ("target", "positive", "uncond", "neutral")), settings)
    cat = lambda n: train_util.concat_embeddings(to(emb["uncond"]), to(emb[n]), 1)
    pcat = lambda n: train_util.concat_embeddings(to(pool["uncond"]), to(pool[n]), 1)
    time_ids = train_util.get_add_time_ids(hw * 8, hw * 8, dtype=torch.bfloat16).to(dev) if cfg.is_xl else None
    tids2 = train_util.concat_embeddings(time_ids, time_ids, 1) if cfg.is_xl else None

    def predict(lat, n, t, gscale):
        if cfg.is_xl:
            return train_util.predict_noise_xl(unet, sched, t, lat, text_embeddings=cat(n), add_text_embeddings=pcat(n),
                                               add_time_ids=tids2, guidance_scale=gscale)
        return train_util.predict_noise(unet, sched, t, lat, cat(n), guidance_scale=gscale)

    with torch.no_grad():
        sched.set_timesteps(50, device=dev)
        optimizer.zero_grad()
        latents = to(noise * sched.init_noise_sigma)
        with network:
            if cfg.is_xl:
                denoised = train_util.diffusion_xl(unet, sched, latents, text_embeddings=cat("target"),
                                                   add_text_embeddings=pcat("target"), add_time_ids=tids2,
                                                   start_timesteps=0, total_timesteps=k, guidance_scale=3)
            else:
                denoised = train_util.diffusion(unet, sched, latents, cat("target"), start_timesteps=0,
                                                total_timesteps=k, guidance_scale=3)
        sched.set_timesteps(1000)
        t_cur = sched.timesteps[int(k * 1000 / 50)]
        positive = predict(denoised, "positive", t_cur, 1)
        neutral = predict(denoised, "neutral", t_cur, 1)
        uncond = predict(denoised, "uncond", t_cur, 1)
    with network:
        target = predict(denoised, "target", t_cur, 1)
    assert target.requires_grad, "the engine output must be differentiable w.r.t. the flat adapter parameter"
    positive.requires_grad = False
    neutral.requires_grad = False
    uncond.requires_grad = False
    loss = pair.loss(target_latents=target, positive_latents=positive, neutral_latents=neutral,
                     unconditional_latents=uncond)
    loss.backward()
    grad_a = network.flat_parameter().grad.detach().float().cpu().clone()
    optimizer.step()
    torch.cuda.synchronize()
    params_a = st.params.detach().clone()
    ckpt = tmp_path / f"seam_alpha1.0_rank4_noxattn_last.pt"
    network.save_weights(ckpt, dtype=torch.bfloat16)

    # the checkpoint loads STRICT into the reference-shaped network (what the inference notebooks do)
    nw = LoRANetworkOracle(build_unet(name, seed=0), rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    missing = nw.load_state_dict(torch.load(ckpt), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys

    # ---------------- B: the fused trainer on the same inputs ----------------
    eng = UNetEngine(cfg, net.state_dict(), dev)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    store.params.copy_(params0)
    tr = SliderTrainer(eng, store, hw, hw, lr=2e-4)
    c2 = lambda n: torch.cat([emb["uncond"], emb[n]]).to(dev, torch.bfloat16).contiguous()
    p2 = (lambda n: torch.cat([pool["uncond"], pool[n]]).to(dev, torch.bfloat16).contiguous()) if cfg.is_xl else (lambda n: None)
    pe = PairEmbeds(c2("target"), c2("positive"), c2("neutral"), c2("uncond"), p2("target"), p2("positive"),
                    p2("neutral"), p2("uncond"), guidance_scale=gs, action=action)
    loss_b = tr.iteration(pe, k, noise.to(dev))
    torch.cuda.synchronize()

    r_den = rel_err(denoised.float().cpu(), tr.denoised.float().cpu())
    r_tgt = rel_err(target.detach().float().cpu(), tr.e_tgt.float().cpu())
    grad_b = store.grads.float().cpu()
    cos = F.cosine_similarity(grad_a, grad_b, dim=0).item()
    dpa = (params_a.float() - params0.float()).cpu()
    dpb = (store.params.float() - params0.float()).cpu()
    agree = (torch.sign(dpa) == torch.sign(dpb)).float().mean().item()
    print(f"[seam] {name}: denoised rel_l2 {r_den:.3e}, target eps rel_l2 {r_tgt:.3e}, loss {loss.item():.5e} vs fused "
          f"{loss_b.item():.5e}, grad cosine {cos:.6f}, |g| {grad_a.norm():.3e} vs {grad_b.norm():.3e}, "
          f"update sign agreement {agree:.4f}, max |dparam| {dpa.abs().max():.2e} / {dpb.abs().max():.2e}")
    # both paths replay the same command buffers on the same inputs and the engine is bit-reproducible (round 3: fixed-order
    # reductions), so the denoised latents and the target prediction are EQUAL; what differs is the glue around them - the
    # loss and its gradient computed by torch bf16 ops here vs the fused fp32 kernel, three frozen CFG pairs vs the
    # de-duplicated B=3 pass (measured: loss 0.16 - 0.31 % apart, gradient cosine 1.0000)
    assert r_den == 0.0 and r_tgt == 0.0
    assert abs(loss.item() - loss_b.item()) < 0.01 * abs(loss_b.item())
    assert cos > 0.9995
    assert agree > 0.85
    assert abs(grad_a.norm().item() / grad_b.norm().item() - 1.0) < 0.01
    assert 0 < dpa.abs().max() < 5e-4


def test_integration_md_c_abi_stub_runs_as_written(dev):
    """the ctypes stub a maintainer of the reference would write, executed VERBATIM from the
    document (so the documented struct cannot drift from include/sliders_hip.h): one fused LoRAModule.forward launch."""
    import ctypes as C
    import math
    import os
    import re
    from sliders_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "class GemmDesc(C.Structure)" in b)
    block = block.replace('C.CDLL("sliders_amd/libsliders_hip.so")', f'C.CDLL("{lib.LIB_PATH}")')
    torch.manual_seed(0)
    M, N, K = 200, 192, 256
    bf = lambda t: t.to(device=dev, dtype=torch.bfloat16).contiguous()
    ns = dict(x=bf(torch.randn(M, K)), W=bf(torch.randn(N, K) / math.sqrt(K)), b=bf(torch.randn(N)),
              y=torch.zeros(M, N, device=dev, dtype=torch.bfloat16), A=bf(torch.randn(4, K) / math.sqrt(K)),
              B=bf(torch.randn(N, 4)), s=torch.tensor([0.25], device=dev), M=M, N=N, K=K)
    lib.load()
    exec(block, ns)
    torch.cuda.synchronize()
    assert C.sizeof(ns["GemmDesc"]) == C.sizeof(lib.GemmDesc)
    x, W, b, A, B = (ns[k].float() for k in ("x", "W", "b", "A", "B"))
    ref = x @ W.t() + b + 0.25 * (x @ A.t()).to(torch.bfloat16).float() @ B.t()
    assert rel_err(ns["y"].float().cpu(), ref.cpu()) < 6e-3
