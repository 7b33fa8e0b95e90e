"""CPU tests that pin the oracle (and the product-side host logic) to the reference.

The golden files under tests/golden/ were produced by the reference's own unmodified Python in the authoring
container (tests/golden/make_golden.py); nothing here reads /root/reference.
"""
import hashlib
import json
import math
import os

import pytest
import torch

from oracle.ddim_oracle import DDIMScheduler
from oracle.lora_oracle import LoRANetworkOracle
from oracle.unet_oracle import build_unet
from sliders_amd.config import CONFIGS
from sliders_amd.ddim import DDIMSchedule
from sliders_amd.lora_store import LoraStore
from sliders_amd.modules import build_tree, lora_targets

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_param_counts_match_published_architectures():
    """859,520,964 (SD-1.x), 865,910,724 (SD-2.x) and 2,567,463,684 (SDXL-base) UNet parameters."""
    for name, n in (("sd1", 859_520_964), ("sd2", 865_910_724), ("sdxl", 2_567_463_684)):
        net = build_unet(name, device="meta")
        assert sum(p.numel() for p in net.parameters()) == n


def test_module_tree_matches_oracle():
    for name in ("sd1", "sd2", "sdxl", "tiny_sd1", "tiny_sd2", "tiny_sdxl"):
        net = build_unet(name, device="meta")
        a = [(n, m.__class__.__name__) for n, m in net.named_modules()]
        b = [(n, m.cls) for n, m in build_tree(CONFIGS[name]()).named_modules()]
        assert a == b


def _key_shapes(store):
    return [[k, list(v.shape)] for k, v in store.state_dict().items()]


@pytest.mark.parametrize("name", ["sd1", "sd2", "sdxl"])
def test_lora_census_and_checkpoint_layout_match_reference(name):
    """Key order, names and shapes of LoRANetwork.state_dict() (lora.py:231-248) - what the reference's
    inference notebooks strict-load."""
    gold = json.load(open(os.path.join(G, "lora_census.json")))
    for method in ("noxattn", "full", "xattn", "xattn-strict", "selfattn", "innoxattn", "noxattn-hspace", "noxattn-hspace-last"):
        ent = gold[f"{name}/{method}"]
        tg = lora_targets(CONFIGS[name](), method)
        assert len(tg) == ent["modules"]
        keys = []
        n_params = 0
        for t in tg:
            k = 3 if t.kind == "conv3" else 1
            dn = [t.rank, t.in_dim] if t.kind == "linear" else [t.rank, t.in_dim, k, k]
            up = [t.out_dim, t.rank] if t.kind == "linear" else [t.out_dim, t.rank, 1, 1]
            keys += [[t.lora_name + ".alpha", []], [t.lora_name + ".lora_down.weight", dn],
                     [t.lora_name + ".lora_up.weight", up]]
            n_params += t.rank * t.in_dim * (9 if t.kind == "conv3" else 1) + t.out_dim * t.rank
        assert n_params == ent["params"]
        text = "\n".join(f"{k}:{tuple(s)}" for k, s in keys)
        assert hashlib.sha256(text.encode()).hexdigest() == ent["sha256"], (name, method)
        if "keys" in ent:
            assert keys == ent["keys"]
    assert gold["sd1/noxattn"]["modules"] == 150 and gold["sdxl/noxattn"]["modules"] == 346


def test_lora_store_state_dict_roundtrip_and_layout():
    cfg = CONFIGS["tiny_sdxl"]()
    torch.manual_seed(3)
    s = LoraStore(cfg, train_method="full")
    s.params.copy_(torch.randn(s.numel).to(torch.bfloat16))
    sd = s.state_dict()
    s2 = LoraStore(cfg, train_method="full", init="none")
    s2.load_state_dict(sd)
    assert torch.equal(s.params, s2.params)
    # fused groups are adjacent in the packed buffer
    grp = s.fused_group([f"down_blocks.1.attentions.0.transformer_blocks.0.attn1.{x}" for x in ("to_q", "to_k", "to_v")])
    assert grp is not None and grp[1].down_off == grp[0].down_off + grp[0].down_numel
    with pytest.raises(KeyError):
        s2.load_state_dict({k: v for k, v in list(sd.items())[3:]})


def test_lora_oracle_matches_reference_golden():
    """oracle/lora_oracle.py + oracle UNet reproduce what the reference's LoRANetwork / predict_noise[_xl] /
    diffusion[_xl] computed (fp32) in the authoring container."""
    gold = torch.load(os.path.join(G, "tiny_forward.pt"))
    for key, e in gold.items():
        name, method = key.split("/")
        net = build_unet(name, seed=0)
        nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
        nw.load_state_dict(e["lora_state_dict"], strict=True)
        lat, ctx, t = e["latents"], e["ctx"], e["t"]
        kw = {"text_embeds": e["pooled"], "time_ids": e["time_ids"]} if "pooled" in e else None
        sch = DDIMScheduler()
        sch.set_timesteps(50)
        with torch.no_grad():
            with nw:
                eps_on = net(torch.cat([lat] * 2), torch.tensor(t), ctx, kw).sample
                u, c = eps_on.chunk(2)
                pred = u + 3 * (c - u)
                x = lat
                for ts in sch.timesteps[0:3]:
                    ep = net(torch.cat([x] * 2), ts, ctx, kw).sample
                    uu, cc = ep.chunk(2)
                    x = sch.step(uu + 3 * (cc - uu), ts, x).prev_sample
            eps_off = net(torch.cat([lat] * 2), torch.tensor(t), ctx, kw).sample
        for got, ref, nm in ((eps_on, e["eps_on"], "eps_on"), (eps_off, e["eps_off"], "eps_off"),
                             (pred, e["pred_on_g3"], "pred"), (x, e["denoised_3"], "denoised")):
            err = (got - ref).abs().max().item()
            assert err < 2e-5, f"{key} {nm}: {err}"


def test_loss_golden():
    gold = torch.load(os.path.join(G, "loss.pt"))
    t = gold["inputs"]
    for action, sign in (("erase", -1), ("enhance", 1)):
        tl = t["target"].clone().requires_grad_(True)
        y = t["neutral"] + sign * 4.0 * (t["positive"] - t["unconditional"]) if sign == 1 else \
            t["neutral"] - 4.0 * (t["positive"] - t["unconditional"])
        l = torch.nn.functional.mse_loss(tl, y)
        l.backward()
        assert torch.equal(l.detach(), gold[action]["loss"])
        assert torch.equal(tl.grad, gold[action]["grad"])


def test_ddim_closed_form_and_tables():
    sch = DDIMScheduler()
    sch.set_timesteps(50)
    assert sch.timesteps[:3].tolist() == [980, 960, 940] and sch.timesteps[-1].item() == 0
    sch.set_timesteps(1000)
    # reference quirk D.1: epsilon is evaluated at timesteps_1000[int(k*1000/50)] = 999 - 20k
    assert [int(sch.timesteps[int(k * 1000 / 50)]) for k in (1, 25, 49)] == [979, 499, 19]
    sch.set_timesteps(50)
    a = torch.cumprod(1 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2, 0)
    x = torch.randn(1, 4, 8, 8, dtype=torch.float64)
    e = torch.randn(1, 4, 8, 8, dtype=torch.float64)
    for t in (980, 500, 20, 0):
        at, ap = a[t], (a[t - 20] if t >= 20 else torch.tensor(1.0, dtype=torch.float64))
        x0 = (x - (1 - at).sqrt() * e) / at.sqrt()
        ref = ap.sqrt() * x0 + (1 - ap).sqrt() * e
        got = sch.step(e.float(), t, x.float()).prev_sample
        assert (got.double() - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())
        # product-side scalar table agrees with the oracle's
        p = DDIMSchedule()
        assert p.step_coefficients(t, 50) == sch.step_coefficients(t)
    assert DDIMSchedule().make_timesteps(50) == sch.timesteps.tolist()


def test_ddim_v_prediction_closed_form():
    """pretrained_model.v_pred (model_util.py:126): x0 = sqrt(a) x - sqrt(1-a) v, eps = sqrt(a) v + sqrt(1-a) x, then the same
    eta = 0 update; oracle and product-side coefficient tables agree."""
    sch = DDIMScheduler(prediction_type="v_prediction")
    sch.set_timesteps(50)
    a = torch.cumprod(1 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2, 0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 8, 8, dtype=torch.float64, generator=g)
    v = torch.randn(1, 4, 8, 8, dtype=torch.float64, generator=g)
    p = DDIMSchedule(prediction_type="v_prediction")
    for t in (980, 500, 20, 0):
        at, ap = a[t], (a[t - 20] if t >= 20 else torch.tensor(1.0, dtype=torch.float64))
        x0 = at.sqrt() * x - (1 - at).sqrt() * v
        e = at.sqrt() * v + (1 - at).sqrt() * x
        ref = ap.sqrt() * x0 + (1 - ap).sqrt() * e
        got = sch.step(v.float(), t, x.float()).prev_sample
        assert (got.double() - ref).abs().max() < 5e-5 * max(1.0, ref.abs().max().item())
        f = p.step_fields(t, 50)
        assert f["v_prediction"] == 1 and f["c_sqrt_alpha_t"] == float(sch.alphas_cumprod[t] ** 0.5)
        assert (f["c_sqrt_beta_t"], f["c_inv_sqrt_alpha_t"], f["c_sqrt_alpha_prev"], f["c_dir"]) == sch.step_coefficients(t)
    assert DDIMSchedule().step_fields(500, 50)["v_prediction"] == 0
    with pytest.raises(ValueError):
        DDIMSchedule(prediction_type="sample")


@pytest.mark.parametrize("name,method", [("tiny_sdxl", "noxattn"), ("tiny_sdxl", "full"), ("tiny_sd1", "noxattn"), ("tiny_sd1", "full")])
def test_lora_init_follows_reference_rng_order(name, method):
    """torch.manual_seed(s) gives the REFERENCE's initial adapter weights: LoraStore.init_reference and the oracle network
    both draw the RNG like lora.py:68-97 / 206-216 (default init of down and up, then kaiming_uniform(a=1); the duplicate
    visits of the conv leaves under DownBlock2D / UpBlock2D are constructed and discarded).  Golden: the reference's own
    LoRANetwork (tests/golden/make_golden.py::lora_init)."""
    gold = json.load(open(os.path.join(G, "lora_init.json")))[f"{name}/{method}"]
    cfg = CONFIGS[name]()
    torch.manual_seed(1234)
    s = LoraStore(cfg, train_method=method)
    sd = s.state_dict()
    net = build_unet(name, seed=0)
    torch.manual_seed(1234)
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
    assert [m.lora_name for m in nw.unet_loras] == list(gold.keys()) or set(m.lora_name for m in nw.unet_loras) == set(gold)
    for m in nw.unet_loras:
        ref = m.lora_down.weight.detach().to(torch.bfloat16)
        g = gold[m.lora_name]
        assert [float(ref.float().sum()), float(ref.float().flatten()[0]), float(ref.float().flatten()[-1])] == g, m.lora_name
        assert torch.equal(sd[m.lora_name + ".lora_down.weight"], ref), m.lora_name
        assert sd[m.lora_name + ".lora_up.weight"].abs().max() == 0


def test_lion_oracle_follows_the_published_algorithm():
    """oracle/optim_oracle.Lion (restating lion_pytorch 0.1.2, requirements.txt:5) against a float64 evaluation of Alg. 1 of
    the Lion paper: c = b1 m + (1-b1) g; theta <- theta (1 - lr wd) - lr sign(c); m <- b2 m + (1-b2) g."""
    from oracle.optim_oracle import Lion
    g_ = torch.Generator().manual_seed(21)
    p = torch.nn.Parameter(torch.randn(4096, generator=g_, dtype=torch.float64))
    opt = Lion([p], lr=1e-3, betas=(0.9, 0.99), weight_decay=0.1)
    th, m = p.detach().clone(), torch.zeros(4096, dtype=torch.float64)
    for _ in range(5):
        g = torch.randn(4096, generator=g_, dtype=torch.float64)
        p.grad = g.clone()
        opt.step()
        th = th * (1 - 1e-3 * 0.1) - 1e-3 * torch.sign(0.9 * m + 0.1 * g)
        m = 0.99 * m + 0.01 * g
        assert torch.allclose(p.detach(), th, rtol=0, atol=1e-12)
    assert torch.allclose(opt.state[id(p)]["exp_avg"], m, rtol=0, atol=1e-12)


def test_vae_oracle_is_pinned_to_the_published_architecture():
    """The SD VAE (AutoencoderKL, block_out_channels (128,256,512,512), 2 layers per block, one 512-channel attention head
    in each mid block) has 83,653,863 parameters - the figure published for sd-vae / the SDXL VAE - and its encoder half
    uses exactly the diffusers key names sliders_amd.vae.VaeEncoder loads."""
    from oracle.vae_oracle import build_vae
    from sliders_amd.vae import random_vae_state_dict
    vae = build_vae("sdxl", with_decoder=True)
    assert sum(p.numel() for p in vae.parameters()) == 83_653_863
    enc_keys = {k for k in vae.state_dict() if k.startswith("encoder.") or k.startswith("quant_conv.")}
    sd = random_vae_state_dict()
    assert set(sd) == enc_keys
    ref = vae.state_dict()
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in sd)
    assert "encoder.mid_block.attentions.0.to_out.0.weight" in sd and "encoder.down_blocks.1.resnets.0.conv_shortcut.weight" in sd
    # forward shape + the posterior algebra of get_noisy_image
    import torch
    from oracle.vae_oracle import AutoencoderKL, get_noisy_image
    small = AutoencoderKL((32, 32, 64, 64), 0.13025, with_decoder=False).eval()
    x = torch.rand(1, 3, 32, 32) * 2 - 1
    dist = small.encode(x).latent_dist
    assert dist.mean.shape == (1, 4, 4, 4) and float(dist.logvar.max()) <= 20.0
    ac = torch.linspace(0.9999, 0.01, 1000)
    n1, n2 = torch.randn(1, 4, 4, 4), torch.randn(1, 4, 4, 4)
    noisy, _ = get_noisy_image(x, small, ac, 500, n1, n2)
    lat = 0.13025 * (dist.mean + dist.std * n1)
    assert torch.allclose(noisy, ac[500].sqrt() * lat + (1 - ac[500]).sqrt() * n2, atol=1e-6)


# ---- Prodigy (train.optimizer: prodigy, train_util.py:369-372; prodigyopt==1.0 absent -> parity unpinned) ------------
def _quadratic(n=64, seed=0):
    import numpy as np
    g = np.random.default_rng(seed)
    return g.uniform(0.5, 2.0, n), g.standard_normal(n) * 3, g.standard_normal(n)


@pytest.mark.parametrize("kw", [dict(), dict(weight_decay=0.01), dict(use_bias_correction=True, safeguard_warmup=True),
                                dict(d_coef=2.0, growth_rate=1.5, betas=(0.8, 0.99))])
def test_prodigy_matches_float64_oracle_and_estimates_the_distance(kw):
    """product (torch ops, float64 parameter here) against the independent numpy restatement over 200 steps of a
    quadratic; plus what the method promises: d never decreases, stays below D = |x0 - x*|, and the iterates converge
    without any learning rate being tuned (lr = 1)"""
    import numpy as np
    from oracle.optim_oracle import ProdigyF64
    from sliders_amd.optim import Prodigy
    a, t, x0 = _quadratic()
    p = torch.tensor(x0, dtype=torch.float64)
    opt, orc = Prodigy([p], lr=1.0, **kw), ProdigyF64(x0, lr=1.0, **kw)
    ds = []
    for _ in range(200):
        p.grad = torch.tensor(a) * (p - torch.tensor(t))
        opt.step()
        orc.step(a * (orc.x - t))
        ds.append(opt.param_groups[0]["d"])
    np.testing.assert_allclose(p.numpy(), orc.x, rtol=1e-4, atol=1e-4)
    assert abs(ds[-1] - orc.d) < 1e-5 * orc.d
    D = float(np.linalg.norm(x0 - t))
    assert all(b >= a_ for a_, b in zip(ds, ds[1:])) and ds[0] <= ds[-1] <= kw.get("d_coef", 1.0) * D
    assert ds[-1] > 1e-2 * D                              # grew by orders of magnitude from d0 = 1e-6
    assert float((p - torch.tensor(t)).norm()) < 0.05 * D


def test_prodigy_bf16_flat_buffer_like_the_trainer_uses_it():
    """bf16 parameter, bf16 states, fp32-accumulated reductions: stays close to the float64 run and moves every element"""
    import numpy as np
    from oracle.optim_oracle import ProdigyF64
    from sliders_amd.optim import Prodigy
    a, t, x0 = _quadratic(256, 3)
    # like the adapters: half of the buffer starts at exactly zero (lora_up, lora.py:58) - bf16 can represent the first
    # 1e-6-sized steps there, which is what lets the estimate leave d0 at all with bf16 parameters
    x0[::2] = 0.0
    p = torch.tensor(x0, dtype=torch.bfloat16)
    x0r = p.double().numpy().copy()
    opt, orc = Prodigy([p], lr=1.0), ProdigyF64(x0r, lr=1.0)
    for _ in range(60):
        g = a * (p.double().numpy() - t)
        p.grad = torch.tensor(g).to(torch.bfloat16)
        opt.step()
        orc.step(a * (orc.x - t))
    st = opt.state[p]
    assert all(st[k].dtype == torch.bfloat16 for k in ("s", "p0", "exp_avg", "exp_avg_sq"))
    assert 0.5 < opt.param_groups[0]["d"] / orc.d < 2.0
    rel = np.linalg.norm(p.double().numpy() - orc.x) / np.linalg.norm(orc.x)
    assert rel < 0.1, rel
    assert (p.double().numpy()[::2] != 0).all()


def test_prodigy_argument_checks_and_get_optimizer():
    from sliders_amd.optim import Prodigy
    from sliders_amd.train_util import get_optimizer
    assert get_optimizer("Prodigy") is Prodigy
    for bad in (dict(d0=0.0), dict(lr=0.0), dict(eps=0.0), dict(betas=(1.0, 0.9))):
        with pytest.raises(ValueError):
            Prodigy([torch.zeros(2)], **bad)
    for name in ("dadaptsgd", "adam8bit"):
        with pytest.raises(ValueError):
            get_optimizer(name)


# ---- D-Adaptation (train.optimizer: dadaptadam / dadaptlion, train_util.py:339-346; dadaptation==3.1 absent -> parity unpinned) ----
@pytest.mark.parametrize("kw", [dict(), dict(weight_decay=0.01), dict(weight_decay=0.01, decouple=True),
                                dict(use_bias_correction=True), dict(growth_rate=1.2, betas=(0.8, 0.99), eps=1e-6)])
def test_dadapt_adam_matches_float64_oracle_and_estimates_the_distance(kw):
    """product (torch ops, float64 parameter) against the independent numpy restatement over 300 steps of a quadratic, plus
    what the method promises: d never decreases, grows by orders of magnitude from d0 without exceeding the distance to the
    solution by more than a small factor, and the iterates converge with lr = 1"""
    import numpy as np
    from oracle.optim_oracle import DAdaptAdamF64
    from sliders_amd.optim import DAdaptAdam
    a, t, x0 = _quadratic()
    p = torch.tensor(x0, dtype=torch.float64)
    opt, orc = DAdaptAdam([p], lr=1.0, **kw), DAdaptAdamF64(x0, lr=1.0, **kw)
    ds = []
    for _ in range(300):
        p.grad = torch.tensor(a) * (p - torch.tensor(t))
        opt.step()
        orc.step(a * (orc.x - t))
        ds.append(opt.param_groups[0]["d"])
    np.testing.assert_allclose(p.numpy(), orc.x, rtol=1e-6, atol=1e-6)
    assert abs(ds[-1] - orc.d) < 1e-8 * orc.d and opt.param_groups[0]["k"] == 300 == opt.state[p]["step"]
    D = float(np.linalg.norm(x0 - t, ord=np.inf))       # Adam's geometry: the estimate tracks the l-infinity distance
    assert all(b >= a_ for a_, b in zip(ds, ds[1:])) and 1e-2 * D < ds[-1] < 10 * D
    if "growth_rate" in kw:
        assert all(b <= a_ * kw["growth_rate"] * (1 + 1e-12) for a_, b in zip(ds, ds[1:]))
    if not kw.get("weight_decay"):
        assert float((p - torch.tensor(t)).norm()) < 0.2 * float(np.linalg.norm(x0 - t))


@pytest.mark.parametrize("kw", [dict(), dict(weight_decay=0.01), dict(betas=(0.95, 0.98))])
def test_dadapt_lion_matches_float64_oracle(kw):
    import numpy as np
    from oracle.optim_oracle import DAdaptLionF64
    from sliders_amd.optim import DAdaptLion
    a, t, x0 = _quadratic()
    p = torch.tensor(x0, dtype=torch.float64)
    opt, orc = DAdaptLion([p], lr=1.0, **kw), DAdaptLionF64(x0, lr=1.0, **kw)
    ds = []
    for _ in range(300):
        p.grad = torch.tensor(a) * (p - torch.tensor(t))
        opt.step()
        orc.step(a * (orc.x - t))
        ds.append(opt.param_groups[0]["d"])
    np.testing.assert_allclose(p.numpy(), orc.x, rtol=1e-7, atol=1e-7)
    assert abs(ds[-1] - orc.d) < 1e-8 * orc.d
    assert all(b >= a_ for a_, b in zip(ds, ds[1:])) and ds[-1] > 1e3 * 1e-6
    # the loss went down without any learning rate being tuned
    f0, f1 = float((a * (x0 - t) ** 2).sum()), float((a * (p.numpy() - t) ** 2).sum())
    assert f1 < 0.5 * f0


def test_dadapt_bf16_flat_buffer_and_interface():
    """bf16 parameter / states as the trainer uses them (fp32-accumulated reductions), two parameter groups sharing one d, the
    package's argument checks, and get_optimizer's names (train_util.py:339-349)"""
    import numpy as np
    from oracle.optim_oracle import DAdaptAdamF64
    from sliders_amd.optim import DAdaptAdam, DAdaptLion
    from sliders_amd.train_util import get_optimizer
    assert get_optimizer("DAdaptAdam") is DAdaptAdam and get_optimizer("dadaptlion") is DAdaptLion
    a, t, x0 = _quadratic(256, 3)
    x0[::2] = 0.0
    p = torch.tensor(x0, dtype=torch.bfloat16)
    orc = DAdaptAdamF64(p.double().numpy().copy(), lr=1.0)
    opt = DAdaptAdam([p], lr=1.0)
    for _ in range(60):
        p.grad = torch.tensor(a * (p.double().numpy() - t)).to(torch.bfloat16)
        opt.step()
        orc.step(a * (orc.x - t))
    assert all(opt.state[p][k].dtype == torch.bfloat16 for k in ("s", "exp_avg", "exp_avg_sq"))
    assert 0.2 < opt.param_groups[0]["d"] / orc.d < 5.0 and torch.isfinite(p.float()).all()
    q1, q2 = torch.randn(8, dtype=torch.float64), torch.randn(8, dtype=torch.float64)
    for cls in (DAdaptAdam, DAdaptLion):
        o = cls([{"params": [q1]}, {"params": [q2], "lr": 0.0}], lr=1.0)
        q2_before = q2.clone()
        q1.grad, q2.grad = torch.randn(8, dtype=torch.float64), torch.randn(8, dtype=torch.float64)
        o.step(); o.step()
        assert torch.equal(q2, q2_before) and o.param_groups[0]["d"] == o.param_groups[1]["d"]
        o.param_groups[1]["lr"] = 0.5
        with pytest.raises(RuntimeError):
            o.step()
        for bad in (dict(d0=0.0), dict(lr=0.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0))):
            with pytest.raises(ValueError):
                cls([torch.zeros(2)], **bad)
        with pytest.raises(NotImplementedError):
            cls([torch.zeros(2)], fsdp_in_use=True)
    with pytest.raises(ValueError):
        DAdaptAdam([torch.zeros(2)], eps=0.0)


@pytest.mark.parametrize("name", ["tiny_sdxl", "tiny_sd1"])
@pytest.mark.parametrize("method,ntype", [("noxattn", "c3lier"), ("full", "c3lier"), ("noxattn", "lierla")])
def test_image_slider_network_order_and_seeded_init(name, method, ntype):
    """The image sliders build their network with trainscripts/imagesliders/lora.py: conv target list WITHOUT DownBlock2D /
    UpBlock2D (no duplicate visits, so one RNG draw per leaf), kaiming a = sqrt(5).  Golden: that file's own LoRANetwork
    under torch.manual_seed(1234) (tests/golden/make_golden.py::lora_init_image); here: what cli_image.py constructs."""
    gold = json.load(open(os.path.join(G, "lora_init_image.json")))[f"{name}/{method}/{ntype}"]
    cfg = CONFIGS[name]()
    torch.manual_seed(1234)
    s = LoraStore(cfg, train_method=method, network_type="c3lier-image" if ntype == "c3lier" else "lierla", kaiming_a=5 ** 0.5)
    assert [e.name for e in s.entries] == gold["order"]
    sd = s.state_dict()
    for nm in gold["order"]:
        w = sd[nm + ".lora_down.weight"].float()
        assert [float(w.sum()), float(w.flatten()[0]), float(w.flatten()[-1])] == gold[nm], nm
        assert sd[nm + ".lora_up.weight"].abs().max() == 0
    if ntype == "c3lier":       # same leaves as the text sliders' list, different RNG consumption
        torch.manual_seed(1234)
        t = LoraStore(cfg, train_method=method, network_type="c3lier", kaiming_a=5 ** 0.5)
        assert [e.name for e in t.entries] == gold["order"]
        assert not torch.equal(t.params, s.params)
