"""Step-level parity: one whole training iteration of the fused trainer (sliders_amd/trainer.py) against the same
iteration written with the reference's loop structure (train_lora_xl.py:162-356) on the CPU oracle in fp32:
partial DDIM denoise with adapters on (guidance 3), three frozen predictions, target prediction, guidance loss,
backward.  Also: the de-duplicated frozen pass equals the three separate CFG-pair passes."""
import pytest
import torch
import torch.nn.functional as F

from oracle.ddim_oracle import DDIMScheduler
from oracle.lora_oracle import LoRANetworkOracle
from oracle.unet_oracle import build_unet
from sliders_amd.config import CONFIGS
from sliders_amd.lora_store import LoraStore
from sliders_amd.trainer import PairEmbeds, SliderTrainer
from sliders_amd.unet import UNetEngine
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _setup(dev, name="tiny_sdxl", seed=5):
    cfg = CONFIGS[name]()
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    store = LoraStore(cfg, rank=4, alpha=1.0, train_method="noxattn", device=dev)
    for e in store.entries:
        store.params[e.up_off:e.up_off + e.up_numel] = (torch.randn(e.up_numel, generator=g) * 0.03).to(dev, torch.bfloat16)
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in ("target", "positive", "neutral", "uncond")}
    pool = {k: (torch.randn(1, cfg.pooled_dim, generator=g) if cfg.is_xl else None) for k in emb}
    noise = torch.randn(1, 4, 16, 16, generator=g)
    return cfg, store, emb, pool, noise


def _pair(emb, pool, dev, action="enhance"):
    cat = lambda x: torch.cat([emb["uncond"], x]).to(dev, torch.bfloat16).contiguous()
    pc = lambda x: None if x is None else torch.cat([pool["uncond"], x]).to(dev, torch.bfloat16).contiguous()
    return PairEmbeds(cat(emb["target"]), cat(emb["positive"]), cat(emb["neutral"]), cat(emb["uncond"]),
                      pc(pool["target"]), pc(pool["positive"]), pc(pool["neutral"]), pc(pool["uncond"]),
                      guidance_scale=4.0, action=action)


@pytest.mark.parametrize("name,action,pred", [("tiny_sdxl", "enhance", "epsilon"), ("tiny_sd1", "erase", "epsilon"),
                                              ("tiny_sd2", "enhance", "v_prediction")])
def test_iteration_matches_reference_loop_on_oracle(dev, name, action, pred):
    """SDXL loop (train_lora_xl.py) with an `enhance` pair, SD-1.x loop (train_lora.py) with an `erase` pair, and the
    SD-2.x 768-v setting (pretrained_model.v2 + v_pred: model_util.py:107-128) with the v-prediction DDIM step."""
    k, hw = 3, 16
    cfg, store, emb, pool, noise = _setup(dev, name)
    sd = store.state_dict()
    params0 = store.params.clone()
    eng = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)
    tr = SliderTrainer(eng, store, hw, hw, lr=2e-4, prediction_type=pred)
    loss = tr.iteration(_pair(emb, pool, dev, action), k, noise.to(dev))
    torch.cuda.synchronize()
    assert tr.unet_passes == k + 4
    # ---- the reference loop on the oracle (fp32) ----
    net = build_unet(name, seed=0)
    nw = LoRANetworkOracle(net, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    nw.load_state_dict(sd, strict=True)
    for p_ in nw.parameters():
        p_.requires_grad_(True)
    sch = DDIMScheduler(prediction_type=pred)
    tid = torch.tensor([[128.0, 128.0, 0, 0, 128.0, 128.0]] * 2)

    def predict(x, which, t, g):
        ctx = torch.cat([emb["uncond"], emb[which]])
        kw = {"text_embeds": torch.cat([pool["uncond"], pool[which]]), "time_ids": tid} if cfg.is_xl else None
        e = net(torch.cat([x] * 2), t, ctx, kw).sample
        u, c = e.chunk(2)
        return u + g * (c - u)

    with torch.no_grad():
        sch.set_timesteps(50)
        x = noise.clone()
        with nw:
            for t in sch.timesteps[0:k]:
                x = sch.step(predict(x, "target", t, 3), t, x).prev_sample
        sch.set_timesteps(1000)
        t_cur = sch.timesteps[int(k * 1000 / 50)]
        pos, neu, unc = (predict(x, w, t_cur, 1) for w in ("positive", "neutral", "uncond"))
    with nw:
        tgt = predict(x, "target", t_cur, 1)
    sign = 1.0 if action == "enhance" else -1.0          # prompt_util.py:116-135
    ref_loss = F.mse_loss(tgt, neu + sign * 4.0 * (pos - unc))
    ref_loss.backward()
    flat = torch.zeros(store.numel)
    mods = {m.lora_name: m for m in nw.unet_loras}
    for e in store.entries:
        m = mods[e.name]
        flat[e.down_off:e.down_off + e.down_numel] = store._down_to_kernel(e, m.lora_down.weight.grad)
        flat[e.up_off:e.up_off + e.up_numel] = m.lora_up.weight.grad.reshape(-1)
    r_den = rel_err(tr.denoised.float().cpu(), x)
    r_tgt = rel_err(tr.e_tgt.float().cpu(), tgt.detach())
    cos = F.cosine_similarity(store.grads.cpu(), flat, dim=0).item()
    print(f"[parity] iteration: denoised rel_l2={r_den:.3e} target-eps rel_l2={r_tgt:.3e} loss {loss.item():.5e} vs "
          f"{ref_loss.item():.5e} grad cosine {cos:.5f} |g| {store.grads.norm().item():.3e} vs {flat.norm().item():.3e}")
    # engine (bf16 chain, k denoise steps) vs the reference loop on the fp32 oracle.  The engine is bit-reproducible since
    # round 3, so these are fixed numbers, not samples: denoised 6.2e-3 - 7.6e-3, target eps 1.0e-2 - 1.5e-2 (the
    # torch-bf16 arm of a SINGLE pass already sits at 1.0e-2 - 1.5e-2 on these nets, tests/test_unet_gpu.py), loss within
    # 0.6 % / 4.0 % / 6.3 % (a difference of four such predictions), gradient cosine 0.9966 / 0.9844 / 0.9849
    assert r_den < 1.0e-2 and r_tgt < 2.0e-2
    assert abs(loss.item() - ref_loss.item()) < 0.08 * ref_loss.item()
    assert cos > 0.975
    # AdamW moved every trainable parameter by about lr (first step: |update| ~ lr)
    delta = (store.params.float() - params0.float()).abs()
    assert delta.max().item() < 5e-4 and delta.max().item() > 0


@pytest.mark.parametrize("name", ["tiny_sdxl", "tiny_sd1"])
def test_iteration_is_bit_reproducible_including_gradients(dev, name):
    """Two fresh engines + trainers, the same inputs: loss, LoRA gradient buffer and updated parameters are EQUAL bit for bit.
    The forward pass has been since round 3 (fixed-order GroupNorm / split-K reductions); round 4 removed the last fp32
    atomics of the path, the M-split sums of the weight gradients (slh_wgrad_desc.slabs / tickets), so `loss.backward()`
    (train_lora_xl.py:345) is reproducible too - and data-parallel replicas can be compared by equality, not cosine."""
    k, hw = 3, 16
    runs = []
    for rep in range(2):
        cfg, store, emb, pool, noise = _setup(dev, name)
        eng = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)
        tr = SliderTrainer(eng, store, hw, hw, lr=2e-4)
        losses = [tr.iteration(_pair(emb, pool, dev, "enhance"), k, noise.to(dev)).item() for _ in range(2)]   # two optimizer steps
        torch.cuda.synchronize()
        runs.append((losses, store.grads.clone(), store.params.clone(), tr.e_tgt.clone()))
    (l0, g0, p0, e0), (l1, g1, p1, e1) = runs
    assert l0 == l1
    assert torch.equal(e0, e1)
    ndiff = int((g0 != g1).sum())
    assert torch.equal(g0, g1), f"{ndiff} of {g0.numel()} gradient elements differ between two identical runs"
    assert torch.equal(p0, p1)


@pytest.mark.parametrize("k", [5, None])
def test_iterations_through_graphs_equal_plain_launches(dev, k, monkeypatch):
    """Six iterations replayed as hipGraphs (every once-per-iteration program is captured at the third and REPLAYED from the
    fourth) against the same six as plain launches: loss, four predictions, gradients and parameters equal bit for bit, with torch
    temporaries and a device-to-host copy between iterations (what a training script does around the step: logging, lr schedule).
    Round-4 regression: the programs' leading zero-fill used to be a hipMemsetAsync node, and its replays on the legacy default
    stream stopped zeroing the right bytes once other blits (torch fill_ / copies) had gone through the same stream in between -
    from the second replay on the frozen predictions drifted and the target prediction / gradients were garbage
    (profiles/r04_graph_memset_node.md).  SLH_OP_MEMSET is a kernel of the library now."""
    from sliders_amd import lib

    def run(graphs):
        monkeypatch.setattr(lib, "_GRAPHS_ON", graphs)
        cfg, store, emb, pool, noise = _setup(dev, "tiny_sdxl")
        eng = UNetEngine(cfg, build_unet("tiny_sdxl", seed=0).state_dict(), dev)
        tr = SliderTrainer(eng, store, 16, 16, lr=2e-4)
        pair = _pair(emb, pool, dev)
        out = []
        for it in range(6):
            loss = tr.iteration(pair, k or 2 + it, noise.to(dev)).item()
            logged = (store.grads * tr.grad_scale).to(torch.bfloat16).cpu()        # temporaries + a blit on the same stream
            out.append((loss, tr.e_pos.clone(), tr.e_neu.clone(), tr.e_unc.clone(), tr.e_tgt.clone(), store.grads.clone(),
                        store.params.clone()))
            assert torch.isfinite(logged.float()).all()
        p_tr = eng.plan(2 * tr.bs, 16, 16, "train")
        assert (p_tr.prog._graphs is not None) == graphs and (p_tr.backward.prog._graphs is not None) == graphs
        return out

    plain, graph = run(False), run(True)
    for it, (a, b) in enumerate(zip(plain, graph)):
        assert a[0] == b[0], f"iteration {it}: loss {b[0]} through graphs, {a[0]} plain"
        for name, x, y in zip(("positive", "neutral", "uncond", "target", "grads", "params"), a[1:], b[1:]):
            assert torch.equal(x, y), f"iteration {it}: {name} differs between graph replay and plain launches"


@pytest.mark.parametrize("name", ["tiny_sdxl", "tiny_sd1"])
def test_step_programs_equal_fill_pass_combine_launches(dev, name):
    """The denoise loop as one program per step (timestep from a device table row, guided combine + DDIM update as the
    program's last op; every step's graph captured after the first loop) against the loop as it was: fill io['t'], replay
    the pass, launch slh_cfg_ddim.  Five iterations with k walking over captured and not-yet-run step indices: denoised
    latents, loss, gradients and parameters equal bit for bit."""
    from sliders_amd import lib

    def run(step_graphs):
        cfg, store, emb, pool, noise = _setup(dev, name)
        eng = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)
        tr = SliderTrainer(eng, store, 16, 16, lr=2e-4)
        tr.step_graphs = step_graphs
        pair = _pair(emb, pool, dev)
        out = []
        for k in (2, 7, 1, 49, 4):
            loss = tr.iteration(pair, k, noise.to(dev)).item()
            out.append((loss, tr.denoised.clone(), store.grads.clone(), store.params.clone()))
        p_on = eng.plan(2 * tr.bs, 16, 16, "on")
        sp = getattr(p_on, "_step_progs", None)
        assert (sp is not None) == step_graphs
        if step_graphs:
            assert sp.captured and all(pr is not None for pr in sp.progs[:49])
            if p_on.prog.n_ops >= lib.Program.GRAPH_MIN_OPS and lib._GRAPHS_ON:
                assert all(pr._graphs is not None for pr in sp.progs[:49])
            assert sp.progs[3].op_names[-1] == "cfg_ddim_step" and sp.progs[3].n_ops == (p_on.prog_text_cached or p_on.prog).n_ops + 1
            assert sp.progs[0].n_ops == p_on.prog.n_ops + 1
        return out

    old, new = run(False), run(True)
    for it, (a, b) in enumerate(zip(old, new)):
        assert a[0] == b[0], f"iteration {it}: loss {b[0]} with step programs, {a[0]} with separate launches"
        for nm, x, y in zip(("denoised", "grads", "params"), a[1:], b[1:]):
            assert torch.equal(x, y), f"iteration {it}: {nm} differs"


def test_dedup_frozen_equals_three_cfg_pairs(dev):
    """Same engine, same denoised latents, same timestep: the one-pass [uncond, positive, neutral] evaluation
    against the reference's three CFG-pair passes.  Every reduction of the pass runs in a fixed order (round 3), rows of different
    samples never meet in one accumulator, and on these nets both batch sizes resolve to the same tiles - so the three predictions
    are EQUAL, bit for bit.  (On shapes where the tuned table picks another split-K factor for M = 3 HW than for 2 HW the K sum is
    re-associated and the results differ in the last fp32 bits of a product: allowed here up to the value measured on MI355X, 0.)"""
    name, k, hw = "tiny_sdxl", 2, 16
    cfg, store, emb, pool, noise = _setup(dev)
    eng = UNetEngine(cfg, build_unet(name, seed=0).state_dict(), dev)
    tr = SliderTrainer(eng, store, hw, hw, dedup_frozen=True)
    pair = _pair(emb, pool, dev)
    tr.iteration(pair, k, noise.to(dev))
    torch.cuda.synchronize()
    ded = [t.float().cpu().clone() for t in (tr.e_pos, tr.e_neu, tr.e_unc)]
    t_cur = tr.t1000[int(k * 1000 / tr.nsteps)]
    eng.set_lora(False)
    p_off = eng.plan(2, hw, hw, "off")
    tr._predict(p_off, tr.denoised, pair.ctx_positive, pair.pooled_positive, t_cur, tr.e_pos)
    tr._predict(p_off, tr.denoised, pair.ctx_neutral, pair.pooled_neutral, t_cur, tr.e_neu)
    tr._predict(p_off, tr.denoised, pair.ctx_uncond, pair.pooled_uncond, t_cur, tr.e_unc)
    torch.cuda.synchronize()
    for a, b, nm in zip(ded, (tr.e_pos, tr.e_neu, tr.e_unc), ("pos", "neu", "unc")):
        b = b.float().cpu()
        r = rel_err(a, b)
        print(f"[parity] dedup {nm}: rel_l2 {r:.3e} max abs diff {(a - b).abs().max().item():.3e}")
        assert torch.equal(a, b), f"dedup {nm}: the B=3 pass and the CFG-pair pass must agree bit for bit on this net (rel_l2 {r:.3e})"
