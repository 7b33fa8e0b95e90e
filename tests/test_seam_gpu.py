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
    # both paths run the same kernels per UNet pass; they differ in where the bf16 roundings of the glue sit (torch ops
    # vs fused kernels), in the de-duplicated frozen pass and in the fp32 atomics order of GroupNorm statistics.  The
    # run-to-run floor of ONE path is already rel_l2 ~7e-3 on the denoised latents / cosine ~0.993 on the gradient
    # (tests/test_rccl_gpu.py prints it), so the bounds are 1.5 x that floor
    assert r_den < 1.2e-2 and r_tgt < 2.0e-2
    # two runs of the SAME iteration differ by up to ~5 % in the loss (a difference of four predictions that each carry the
    # engine's run-to-run floor - GroupNorm statistics are fp32 atomics); measured spread of this comparison: 0.3 - 4.6 %
    assert abs(loss.item() - loss_b.item()) < 0.08 * abs(loss_b.item())
    assert cos > 0.97          # measured 0.9945 (tiny_sdxl), 0.979 (tiny_sd1) against a run-to-run floor of 0.993
    assert agree > 0.85
    assert abs(grad_a.norm().item() / grad_b.norm().item() - 1.0) < 0.08
    assert 0 < dpa.abs().max() < 5e-4
