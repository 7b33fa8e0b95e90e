"""Image-slider path (trainscripts/imagesliders): the fp32 AutoencoderKL encoder kernels, `get_noisy_image` and one whole
image-slider iteration of the HIP engine against the CPU oracle (oracle/vae_oracle.py + the reference loop of
train_lora-scale-xl.py:178-396 written over the oracle UNet).

Tolerances.  The VAE path is fp32 on both sides (the reference keeps the VAE in fp32): relative L2 error < 2e-5 per kernel
and < 1e-4 for the whole encoder (accumulation order of 30 convolutions; GroupNorm statistics by fp32 atomics).  The
noisy latents are handed to the UNet in bf16, bit-identical rounding of the fp32 result.  The iteration is compared like
the text-slider iteration (tests/test_trainer_gpu.py): gradient cosine vs fp32 autograd through the oracle.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle
from oracle.ddim_oracle import DDIMScheduler
from oracle.lora_oracle import LoRANetworkOracle
from oracle.unet_oracle import build_unet
from sliders_amd import lib
from sliders_amd.config import CONFIGS
from sliders_amd.image_trainer import ImageSliderTrainer
from sliders_amd.lora_store import LoraStore
from sliders_amd.trainer import PairEmbeds
from sliders_amd.unet import UNetEngine
from sliders_amd.vae import VaeEncoder, random_vae_state_dict
from tests.util import p, rel_err, report, stream

pytestmark = pytest.mark.gpu


def _pix(x):          # NCHW -> [B*H*W][C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 12, 20, 32, 48), (1, 16, 16, 128, 260), (1, 9, 7, 16, 4)])
def test_sgemm_conv3x3(dev, stride, B, H, W, Ci, Co, split):
    """split = 0: exact-fp32 MFMA; 1: operands split into bf16 hi + lo, three bf16 MFMAs per block (16 mantissa bits; Cin = 16
    falls back to the exact kernel)."""
    torch.manual_seed(Ci + Co + stride)
    img = torch.randn(B, Ci, H, W, device=dev)
    w4 = torch.randn(Co, Ci, 3, 3, device=dev) / math.sqrt(9 * Ci)
    bias = torch.randn(Co, device=dev)
    if stride == 1:
        ref = F.conv2d(img, w4, bias, padding=1)
    else:                       # Downsample2D(padding=0): zero pad right/bottom, then k3 s2
        ref = F.conv2d(F.pad(img, (0, 1, 0, 1)), w4, bias, stride=2)
    Ho, Wo = ref.shape[2:]
    res = torch.randn(B * Ho * Wo, Co, device=dev)
    x = _pix(img)
    wp = w4.permute(0, 2, 3, 1).reshape(Co, -1).contiguous()
    c = torch.zeros(B * Ho * Wo, Co, device=dev)
    d = lib.SgemmDesc(x=p(x), w=p(wp), bias=p(bias), residual=p(res), c=p(c), ldx=Ci, ldw=9 * Ci, ldr=Co, ldc=Co,
                      M=B * Ho * Wo, N=Co, K=9 * Ci, mode=1, cin=Ci, batch=B, hs=H, ws=W, ho=Ho, wo=Wo, stride=stride,
                      pad=1 if stride == 1 else 0, alpha=1.0, split_bf16=split)
    lib.call(lib.OP_SGEMM, d, stream())
    torch.cuda.synchronize()
    report(f"sgemm conv s{stride} {B}x{Ci}x{H}x{W}->{Co} split{split}", c, _pix(ref) + res, 4e-5 if split else 2e-5)


def test_sgemm_dense_alpha_rowbias(dev):
    torch.manual_seed(3)
    M, N, K = 300, 132, 64
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / 8
    bn, bm = torch.randn(N, device=dev), torch.randn(M, device=dev)
    for per_row, split in ((0, 0), (1, 0), (0, 1), (1, 1)):
        c = torch.zeros(M, N, device=dev)
        d = lib.SgemmDesc(x=p(x), w=p(w), bias=p(bm if per_row else bn), c=p(c), ldx=K, ldw=K, ldc=N, M=M, N=N, K=K, mode=0,
                          alpha=0.37, bias_per_row=per_row, split_bf16=split)
        lib.call(lib.OP_SGEMM, d, stream())
        torch.cuda.synchronize()
        ref = 0.37 * (x.double() @ w.double().t()).float() + (bm[:, None] if per_row else bn[None, :])
        report(f"sgemm dense per_row={per_row} split{split}", c, ref, 4e-5 if split else 2e-5)


def test_gn32_and_softmax32(dev):
    torch.manual_seed(4)
    B, HW, C = 2, 300, 128
    gm, bt = torch.randn(C, device=dev), torch.randn(C, device=dev)
    for act, off in ((0, 0.5), (1, 0.5), (1, 300.0)):       # 300: DC offset >> spread, the one-pass variance hazard
        x = torch.randn(B * HW, C, device=dev) * 2 + off
        stats = torch.full((B, 32, 2), float("nan"), device=dev)
        prow, ntick = lib.gn32_workspace(HW)
        part = torch.full((B, prow, 32, 2), float("nan"), device=dev)    # scratch: contents must not matter
        ticket = torch.zeros(B, ntick, dtype=torch.int32, device=dev)
        y = torch.zeros_like(x)
        d = lib.Gn32Desc(x=p(x), gamma=p(gm), beta=p(bt), stats=p(stats), y=p(y), ldx=C, ldy=C, C=C, batch=B, hw=HW, groups=32,
                         eps=1e-6, act=act, partial=p(part), ticket=p(ticket))
        lib.call(lib.OP_GN32_STATS, d, stream())
        lib.call(lib.OP_GN32_APPLY, d, stream())
        torch.cuda.synchronize()
        ref = F.group_norm(x.double().view(B, HW, C).transpose(1, 2), 32, gm.double(), bt.double(), eps=1e-6)
        if act:
            ref = F.silu(ref)
        report(f"gn32 act={act} offset={off}", y, ref.float().transpose(1, 2).reshape(B * HW, C), 2e-5 if off < 1 else 6e-5)
        y0 = y.clone()
        lib.call(lib.OP_GN32_STATS, d, stream())
        lib.call(lib.OP_GN32_APPLY, d, stream())
        torch.cuda.synchronize()
        assert torch.equal(y, y0) and int(ticket.abs().sum()) == 0, "fixed-order reduction: bit-reproducible"
    s = torch.randn(77, 4096, device=dev) * 3
    ref = torch.softmax(s, -1)
    lib.call(lib.OP_SOFTMAX32, lib.Softmax32Desc(x=p(s), ld=4096, rows=77, cols=4096), stream())
    torch.cuda.synchronize()
    report("softmax32", s, ref, 2e-6)


def _oracle_vae(sd, boc, kind="sdxl"):
    vae = vae_oracle.AutoencoderKL(boc, vae_oracle.VAE_SCALING[kind], with_decoder=False)
    missing = vae.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, strict=True)
    return vae.eval()


@pytest.mark.parametrize("boc,B,size,exact", [((128, 128, 256, 256), 2, 64, True), ((128, 128, 256, 256), 2, 64, False),
                                              ((128, 256, 512, 512), 1, 256, True), ((128, 256, 512, 512), 1, 256, False),
                                              # the benchmarked image-slider size (BASELINE configs[4] as bench.py runs it: 512 x 512
                                              # pairs, mid-block attention over T = 4096 tokens), in the default arithmetic
                                              ((128, 256, 512, 512), 1, 512, True)])
def test_vae_encoder_and_get_noisy_image(dev, boc, B, size, exact):
    """sliders_amd.vae.VaeEncoder vs the oracle AutoencoderKL.encode on identical random-init weights, then the whole
    get_noisy_image (train_util.py:200-235) with the two random draws shared.  Both arithmetic modes: exact fp32 products
    (the default; measured 4e-6) and the opt-in bf16 hi/lo split (16 mantissa bits per operand)."""
    sd = random_vae_state_dict(boc, dev, seed=1)
    enc = VaeEncoder(sd, dev, vae_oracle.VAE_SCALING["sdxl"], exact_fp32=exact)
    vae = _oracle_vae(sd, boc)
    g = torch.Generator().manual_seed(2)
    img_u8 = torch.randint(0, 256, (B, size, size, 3), generator=g, dtype=torch.uint8)
    image = torch.cat([VaeEncoder.preprocess(img_u8[b]) for b in range(B)])            # [B][H][W][3]
    assert torch.equal(image[0].permute(2, 0, 1), vae_oracle.preprocess_image(img_u8[0].numpy())[0])
    h = size // 8
    mom = enc.encode_moments(image.to(dev))
    torch.cuda.synchronize()
    with torch.no_grad():
        ref_m = vae.quant_conv(vae.encoder(image.permute(0, 3, 1, 2)))               # [B][8][h][h]
    report(f"vae moments boc{boc} {size}px exact={exact}", mom.cpu(), _pix(ref_m), 2e-5 if exact else 1e-4)
    post, noise = torch.randn(B, 4, h, h, generator=g), torch.randn(B, 4, h, h, generator=g)
    sch = DDIMScheduler()
    t = 980 - 20 * 7                                                                   # timesteps_50[7]
    a = sch.alphas_cumprod[t]
    nb, nf, lat = enc.get_noisy_image(image.to(dev), post.to(dev), noise.to(dev), float(a.sqrt()), float((1 - a).sqrt()))
    torch.cuda.synchronize()
    with torch.no_grad():
        ref_noisy, _ = vae_oracle.get_noisy_image(image.permute(0, 3, 1, 2), vae, sch.alphas_cumprod, t, post, noise)
    report(f"get_noisy_image fp32 exact={exact}", nf.cpu(), ref_noisy, 2e-5 if exact else 1e-4)
    assert torch.equal(nb.cpu(), nf.cpu().to(torch.bfloat16)), "the bf16 latents must be the rounded fp32 ones"


def test_image_slider_iteration_matches_reference_loop_on_oracle(dev):
    """One iteration of train_lora-scale-xl.py:178-396 (both polarities, gradient accumulation, AdamW) against the same
    loop over the oracle UNet + oracle VAE in fp32."""
    name, hw, k, scale = "tiny_sdxl", 16, 5, 2.0
    cfg = CONFIGS[name]()
    boc = (128, 128, 128, 128)
    g = torch.Generator().manual_seed(9)
    torch.manual_seed(9)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev, kaiming_a=5 ** 0.5)
    for e in store.entries:
        store.params[e.up_off:e.up_off + e.up_numel] = (torch.randn(e.up_numel, generator=g) * 0.03).to(dev, torch.bfloat16)
    sd_lora = store.state_dict()
    params0 = store.params.clone()
    emb = {n: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for n in ("positive", "neutral", "uncond")}
    pool = {n: torch.randn(1, cfg.pooled_dim, generator=g) for n in emb}
    size = hw * 8
    img_low = VaeEncoder.preprocess(torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8))
    img_high = VaeEncoder.preprocess(torch.randint(0, 256, (size, size, 3), generator=g, dtype=torch.uint8))
    post, noise = torch.randn(1, 4, hw, hw, generator=g), torch.randn(1, 4, hw, hw, generator=g)
    vsd = random_vae_state_dict(boc, dev, seed=3)
    net = build_unet(name, seed=0)
    eng = UNetEngine(cfg, net.state_dict(), dev)
    enc = VaeEncoder(vsd, dev, vae_oracle.VAE_SCALING["sdxl"])
    tr = ImageSliderTrainer(eng, store, enc, hw, hw, lr=2e-4)
    cat = lambda n: torch.cat([emb["uncond"], emb[n]]).to(dev, torch.bfloat16).contiguous()
    pc = lambda n: torch.cat([pool["uncond"], pool[n]]).to(dev, torch.bfloat16).contiguous()
    pe = PairEmbeds(cat("positive"), cat("positive"), cat("neutral"), cat("uncond"), pc("positive"), pc("positive"),
                    pc("neutral"), pc("uncond"), guidance_scale=1.0, action="enhance")
    lh, ll = tr.iteration(pe, k, img_low.to(dev), img_high.to(dev), scale, post.to(dev), noise.to(dev))
    torch.cuda.synchronize()
    # ---- the reference loop on the oracle (fp32) ----
    vae = _oracle_vae(vsd, boc)
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    nw.load_state_dict(sd_lora, strict=True)
    for q in nw.parameters():
        q.requires_grad_(True)
    sch = DDIMScheduler()
    sch.set_timesteps(50)
    t_add = int(sch.timesteps[k])
    tid = torch.tensor([[size * 1.0, size * 1.0, 0, 0, size * 1.0, size * 1.0]] * 2)
    sch.set_timesteps(1000)
    t_cur = sch.timesteps[int(k * 1000 / 50)]
    losses = []
    for sign, img, which in ((1.0, img_high, "positive"), (-1.0, img_low, "neutral")):
        with torch.no_grad():
            lat, _ = vae_oracle.get_noisy_image(img.permute(0, 3, 1, 2), vae, sch.alphas_cumprod, t_add, post, noise)
            lat = lat.to(torch.bfloat16).float()
        nw.set_lora_slider(sign * scale)
        with nw:
            e = net(torch.cat([lat] * 2), t_cur, torch.cat([emb["uncond"], emb[which]]),
                    {"text_embeds": torch.cat([pool["uncond"], pool[which]]), "time_ids": tid}).sample
        u, c = e.chunk(2)
        loss = F.mse_loss(u + 1.0 * (c - u), noise.to(torch.bfloat16).float())
        loss.backward()
        losses.append(loss.item())
    flat = torch.zeros(store.numel)
    mods = {m.lora_name: m for m in nw.unet_loras}
    for e_ in store.entries:
        m = mods[e_.name]
        flat[e_.down_off:e_.down_off + e_.down_numel] = store._down_to_kernel(e_, m.lora_down.weight.grad)
        flat[e_.up_off:e_.up_off + e_.up_numel] = m.lora_up.weight.grad.reshape(-1)
    cos = F.cosine_similarity(store.grads.cpu(), flat, dim=0).item()
    print(f"[parity] image-slider iteration: loss high {lh.item():.5e} vs {losses[0]:.5e}, low {ll.item():.5e} vs {losses[1]:.5e}, "
          f"grad cosine {cos:.5f} |g| {store.grads.norm().item():.3e} vs {flat.norm().item():.3e}")
    assert abs(lh.item() - losses[0]) < 0.01 * losses[0] and abs(ll.item() - losses[1]) < 0.01 * losses[1]   # measured 0.02 %
    assert cos > 0.995          # measured 0.9976 (a fixed number: the engine is bit-reproducible since round 3)
    delta = (store.params.float() - params0.float()).abs()
    assert 0 < delta.max().item() < 5e-4


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("boc,B,h", [((128, 128, 256, 256), 2, 8), ((128, 256, 512, 512), 1, 16)])
def test_vae_decoder(dev, boc, B, h, exact):
    """sliders_amd.vae.VaeDecoder vs the oracle: vae.decode(latents / scaling_factor) (generate_images_sd1.py:166-168)."""
    from sliders_amd.vae import VaeDecoder
    sd = random_vae_state_dict(boc, dev, seed=4, decoder=True)
    dec = VaeDecoder(sd, dev, vae_oracle.VAE_SCALING["sd1"], exact_fp32=exact)
    vae = vae_oracle.AutoencoderKL(boc, vae_oracle.VAE_SCALING["sd1"], with_decoder=True).eval()
    vae.load_state_dict({k: v.float().cpu() for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(B, 4, h, h, generator=g)
    for dt in (torch.float32, torch.bfloat16):
        z = lat.to(dt)
        img = dec.decode(z.to(dev))
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = vae.decode(z.float() * (1.0 / vae_oracle.VAE_SCALING["sd1"]))           # [B][3][8h][8h]
        report(f"vae decode boc{boc} {dt} exact={exact}", img.cpu(), ref.permute(0, 2, 3, 1), 2e-5 if exact else 1e-4)
        u8 = VaeDecoder.to_uint8(img).cpu()
        ref8 = ((ref.permute(0, 2, 3, 1) / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8)
        assert (u8.int() - ref8.int()).abs().max() <= 1


@pytest.mark.parametrize("name", ["tiny_sdxl", "tiny_sd1"])
def test_slider_sampler_matches_reference_loop_on_oracle(dev, name):
    """The inference loop of eval-scripts/generate_images_sd1.py:160-164 / generate_images_xl.py (start_noise gating of the
    slider scale, CFG 7.5, DDIM) against the same loop over the oracle UNet + oracle LoRA in fp32."""
    from sliders_amd.sampler import SliderSampler
    cfg = CONFIGS[name]()
    hw, steps, start_noise, scale, gs = 16, 6, 600, 2.0, 7.5
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    for e in store.entries:
        store.params[e.up_off:e.up_off + e.up_numel] = (torch.randn(e.up_numel, generator=g) * 0.03).to(dev, torch.bfloat16)
    sd_lora = store.state_dict()
    unc, txt = torch.randn(1, 77, cfg.cross_attention_dim, generator=g), torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
    pool = torch.randn(2, cfg.pooled_dim, generator=g) if cfg.is_xl else None
    noise = torch.randn(1, 4, hw, hw, generator=g)
    net = build_unet(name, seed=0)
    eng = UNetEngine(cfg, net.state_dict(), dev)
    ctx = torch.cat([unc, txt])
    got = SliderSampler(eng, store).sample_latents(ctx.to(dev), noise.to(dev), scale=scale, start_noise=start_noise,
                                                   ddim_steps=steps, guidance_scale=gs,
                                                   pooled=pool.to(dev) if pool is not None else None)
    torch.cuda.synchronize()
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    nw.load_state_dict(sd_lora, strict=True)
    sch = DDIMScheduler()
    sch.set_timesteps(steps)
    size = hw * 8.0
    kw = {"text_embeds": pool, "time_ids": torch.tensor([[size, size, 0, 0, size, size]] * 2)} if cfg.is_xl else None
    x = noise.clone()
    used = []
    with torch.no_grad():
        for t in sch.timesteps:
            nw.set_lora_slider(0 if t > start_noise else scale)
            used.append(float(nw.lora_scale))
            with nw:
                e = net(torch.cat([x] * 2), t, ctx, kw).sample
            u, c = e.chunk(2)
            x = sch.step(u + gs * (c - u), t, x).prev_sample
    assert 0.0 in used and scale in used, "the test must exercise both sides of start_noise"
    r = rel_err(got.float().cpu(), x)
    print(f"[parity] sampler {name}: final latents rel_l2 {r:.3e} after {steps} steps (scales used {used})")
    assert r < 2.5e-2
