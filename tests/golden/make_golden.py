"""Generates the committed golden fixtures by running the REFERENCE'S OWN, UNMODIFIED Python
(/root/reference/trainscripts/textsliders/{lora,prompt_util,config_util,train_util}.py) in the authoring
container, over the oracle UNet (class names / module paths of diffusers) with a name-only `diffusers` stub.

The reference repo has no tests or golden vectors of its own, and it cannot travel to
the GPU box, so these fixtures are how the oracle and the HIP path are pinned to the reference's code:

  lora_census.json      LoRANetwork(...).state_dict() key order + shapes for SD-1.x / SD-2.x / SDXL and every
                        train_method (full lists for noxattn, sha256 for the rest)         <- lora.py:164-248
  tiny_forward.pt       tiny SDXL-/SD1-topology UNets with the reference LoRANetwork attached (non-zero up
                        weights): unet(...) under `with network`, predict_noise[_xl], diffusion[_xl] (3 DDIM
                        steps, guidance 3), all fp32                                        <- lora.py:108-112,
                                                                                              train_util.py:145-294
  loss.pt               PromptEmbedsPair.loss (erase / enhance) on seeded bf16 tensors       <- prompt_util.py:108-148
  lora_init.json        the reference's LoRANetwork built under torch.manual_seed(1234) over the tiny UNets: per module
                        [sum, first, last] of the bf16 lora_down weights (pins the RNG draw ORDER, duplicates of the
                        conv leaves included)                                               <- lora.py:68-97, 206-216
  schema.json           the reference's pydantic parse of tests/golden/{config,prompts}_sample.yaml
                                                                                           <- config_util.py, prompt_util.py
  host_helpers.json     get_random_resolution_in_bucket / get_add_time_ids(dynamic_crops) / get_initial_latents /
                        concat_embeddings under fixed seeds and the lr sequences of get_lr_scheduler
                                                                                           <- train_util.py:20-57,136-141,298-419
  lora_init_image.json  the IMAGE sliders' LoRANetwork (trainscripts/imagesliders/lora.py) under torch.manual_seed(1234):
                        module order + seeded lora_down weights                            <- imagesliders/lora.py:19-25,96,150-216
Run:  python tests/golden/make_golden.py [census|tiny_forward|loss|schema|lora_init|host_helpers|lora_init_image ...]   (needs /root/reference; not run
      on the GPU box)
"""
import contextlib
import hashlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/trainscripts/textsliders"
sys.path.insert(0, os.path.join(ROOT, "oracle", "_stubs"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import lora as reflora  # noqa: E402  (reference)
import prompt_util as refprompt  # noqa: E402  (reference)
import config_util as refconfig  # noqa: E402  (reference)
import train_util as reftrain  # noqa: E402  (reference)

from oracle.ddim_oracle import DDIMScheduler  # noqa: E402
from oracle.unet_oracle import build_unet  # noqa: E402

METHODS = ["noxattn", "full", "xattn", "selfattn", "innoxattn", "xattn-strict", "noxattn-hspace", "noxattn-hspace-last"]


def c3lier():
    reflora.DEFAULT_TARGET_REPLACE[:] = ["Attention"] + reflora.UNET_TARGET_REPLACE_MODULE_CONV


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def census():
    out = {}
    for name in ("sd1", "sd2", "sdxl"):
        for method in METHODS:
            c3lier()
            net = build_unet(name, device="meta")
            nw = quiet(reflora.LoRANetwork, net, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
            keys = [[k, list(v.shape)] for k, v in nw.state_dict().items()]
            text = "\n".join(f"{k}:{tuple(s)}" for k, s in keys)
            ent = {"modules": len(nw.unet_loras), "params": sum(p.numel() for p in nw.parameters()),
                   "sha256": hashlib.sha256(text.encode()).hexdigest()}
            if method == "noxattn":
                ent["keys"] = keys
            out[f"{name}/{method}"] = ent
    with open(os.path.join(HERE, "lora_census.json"), "w") as f:
        json.dump(out, f)
    print("census:", {k: v["modules"] for k, v in out.items()})


def tiny_forward():
    res = {}
    for name, method in (("tiny_sdxl", "noxattn"), ("tiny_sd1", "noxattn"), ("tiny_sdxl", "full")):
        c3lier()
        torch.manual_seed(11)
        net = build_unet(name, seed=0)
        cfg = net.cfg
        nw = quiet(reflora.LoRANetwork, net, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
        g = torch.Generator().manual_seed(5)
        for l in nw.unet_loras:
            l.lora_up.weight.data.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.05)
        hw = 16
        lat = torch.randn(1, 4, hw, hw, generator=g)
        emb = [torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for _ in range(2)]   # uncond, target
        ctx = reftrain.concat_embeddings(emb[0], emb[1], 1)
        sch = DDIMScheduler()
        sch.set_timesteps(50)
        entry = {"lora_state_dict": {k: v.clone() for k, v in nw.state_dict().items()}, "latents": lat, "ctx": ctx}
        xl = cfg.addition_embed_type is not None
        if xl:
            pooled = [torch.randn(1, cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim, generator=g)
                      for _ in range(2)]
            ptx = reftrain.concat_embeddings(pooled[0], pooled[1], 1)
            tid = torch.tensor([[128.0, 128.0, 0, 0, 128.0, 128.0]])
            tids = reftrain.concat_embeddings(tid, tid, 1)
            entry.update(pooled=ptx, time_ids=tids)
        with torch.no_grad():
            with nw:
                if xl:
                    entry["eps_on"] = net(torch.cat([lat] * 2), sch.timesteps[3], ctx,
                                          {"text_embeds": ptx, "time_ids": tids}).sample
                    entry["pred_on_g3"] = reftrain.predict_noise_xl(net, sch, sch.timesteps[3], lat, ctx, ptx, tids,
                                                                    guidance_scale=3)
                    entry["denoised_3"] = reftrain.diffusion_xl(net, sch, lat, ctx, ptx, tids, guidance_scale=3,
                                                                total_timesteps=3, start_timesteps=0)
                else:
                    entry["eps_on"] = net(torch.cat([lat] * 2), sch.timesteps[3], ctx).sample
                    entry["pred_on_g3"] = reftrain.predict_noise(net, sch, sch.timesteps[3], lat, ctx, guidance_scale=3)
                    entry["denoised_3"] = reftrain.diffusion(net, sch, lat, ctx, total_timesteps=3, start_timesteps=0,
                                                             guidance_scale=3)
            # outside `with network`: multiplier 0 (lora.py:256-258)
            if xl:
                entry["eps_off"] = net(torch.cat([lat] * 2), sch.timesteps[3], ctx, {"text_embeds": ptx, "time_ids": tids}).sample
            else:
                entry["eps_off"] = net(torch.cat([lat] * 2), sch.timesteps[3], ctx).sample
        entry["t"] = int(sch.timesteps[3])
        res[f"{name}/{method}"] = entry
        print(name, method, "eps_on rms", entry["eps_on"].pow(2).mean().sqrt().item(),
              "on-off", (entry["eps_on"] - entry["eps_off"]).abs().max().item())
    torch.save(res, os.path.join(HERE, "tiny_forward.pt"))


def loss():
    g = torch.Generator().manual_seed(9)
    t = {k: torch.randn(1, 4, 16, 16, generator=g).to(torch.bfloat16) for k in ("target", "positive", "neutral", "unconditional")}
    out = {"inputs": t}
    for action in ("erase", "enhance"):
        st = refprompt.PromptSettings(target="a", positive="b", unconditional="c", neutral="d", action=action,
                                      guidance_scale=4.0)
        pair = refprompt.PromptEmbedsPair(torch.nn.MSELoss(), None, None, None, None, st)
        tl = t["target"].clone().requires_grad_(True)
        l = pair.loss(target_latents=tl, positive_latents=t["positive"], neutral_latents=t["neutral"],
                      unconditional_latents=t["unconditional"])
        l.backward()
        out[action] = {"loss": l.detach(), "grad": tl.grad.clone()}
    torch.save(out, os.path.join(HERE, "loss.pt"))
    print("loss:", {a: float(out[a]["loss"]) for a in ("erase", "enhance")})


def schema():
    cfgp = os.path.join(HERE, "config_sample.yaml")
    prp = os.path.join(HERE, "prompts_sample.yaml")
    cfg = refconfig.load_config_from_yaml(cfgp)
    pr_plain = quiet(refprompt.load_prompts_from_yaml, prp)
    pr_attr = quiet(refprompt.load_prompts_from_yaml, prp, ["male", "female"])
    out = {"config": json.loads(cfg.json()), "prompts": [json.loads(p.json()) for p in pr_plain],
           "prompts_attr": [json.loads(p.json()) for p in pr_attr]}
    with open(os.path.join(HERE, "schema.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("schema: ok", len(pr_plain), len(pr_attr))


def lora_init():
    """Seeded initial adapter weights of the reference's own network (RNG order incl. the duplicate conv visits)."""
    out = {}
    for name in ("tiny_sdxl", "tiny_sd1"):
        for method in ("noxattn", "full"):
            c3lier()
            net = build_unet(name, seed=0)
            torch.manual_seed(1234)
            nw = quiet(reflora.LoRANetwork, net, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
            ent = {}
            for m in nw.unet_loras:
                w = m.lora_down.weight.detach().to(torch.bfloat16).float()
                ent[m.lora_name] = [float(w.sum()), float(w.flatten()[0]), float(w.flatten()[-1])]
            out[f"{name}/{method}"] = ent
    with open(os.path.join(HERE, "lora_init.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


def lora_init_image():
    """The IMAGE sliders' network (trainscripts/imagesliders/lora.py: conv target list without DownBlock2D / UpBlock2D, no
    name de-duplication, kaiming a = sqrt(5)) under torch.manual_seed(1234): module order and seeded initial weights."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("reflora_image", "/root/reference/trainscripts/imagesliders/lora.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name in ("tiny_sdxl", "tiny_sd1"):
        for method, ntype in (("noxattn", "c3lier"), ("full", "c3lier"), ("noxattn", "lierla")):
            # DEFAULT_TARGET_REPLACE aliases UNET_TARGET_REPLACE_MODULE_TRANSFORMER (lora.py:29) and train_lora-scale-xl.py:61-63
            # extends it in place for c3lier: reset to the literal before every build
            mod.DEFAULT_TARGET_REPLACE[:] = ["Attention"]
            if ntype == "c3lier":
                mod.DEFAULT_TARGET_REPLACE += mod.UNET_TARGET_REPLACE_MODULE_CONV
            net = build_unet(name, seed=0)
            torch.manual_seed(1234)
            nw = quiet(mod.LoRANetwork, net, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
            names = [m.lora_name for m in nw.unet_loras]
            assert len(names) == len(set(names))
            ent = {"order": names}
            for m in nw.unet_loras:
                w = m.lora_down.weight.detach().to(torch.bfloat16).float()
                ent[m.lora_name] = [float(w.sum()), float(w.flatten()[0]), float(w.flatten()[-1])]
            out[f"{name}/{method}/{ntype}"] = ent
    with open(os.path.join(HERE, "lora_init_image.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("lora_init_image:", {k: len(v) - 1 for k, v in out.items()})


def host_helpers():
    """The reference's host-side helpers of the loop under fixed seeds (train_util.py:20-57, 136-141, 298-333, 376-419):
    which RNG they consume, in which order, and what they return."""
    import types
    out = {}
    for seed in (0, 1, 7):
        torch.manual_seed(seed)
        out[f"bucket/{seed}"] = [list(reftrain.get_random_resolution_in_bucket(b)) for b in (512, 1024, 512, 768, 1024)]
        torch.manual_seed(seed)
        out[f"time_ids_dynamic/{seed}"] = [reftrain.get_add_time_ids(h, w, dynamic_crops=True).tolist()
                                           for h, w in ((512, 512), (1024, 1024), (768, 512))]
    out["time_ids_static"] = reftrain.get_add_time_ids(1024, 768, dynamic_crops=False).tolist()
    out["time_ids_bf16"] = reftrain.get_add_time_ids(1000, 1001, dtype=torch.bfloat16).float().tolist()
    g = torch.Generator().manual_seed(3)
    sched = types.SimpleNamespace(init_noise_sigma=2.5)
    lat = reftrain.get_initial_latents(sched, 2, 64, 96, 3, generator=g)
    out["initial_latents"] = {"shape": list(lat.shape), "sum": float(lat.double().sum()), "first": float(lat.flatten()[0]),
                              "repeat_equal": bool(torch.equal(lat[:2], lat[2:4]))}
    a, b = torch.arange(6.0).view(1, 2, 3), 10 + torch.arange(6.0).view(1, 2, 3)
    out["concat_embeddings"] = reftrain.concat_embeddings(a, b, 2).tolist()
    # LR schedules (train_util.py:376-404): the lr the optimizer sees at iteration i, lr 2e-4, 1000 iterations
    for name in ("constant", "cosine", "cosine_with_restarts", "step"):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=2e-4)
        sch = reftrain.get_lr_scheduler(name, opt, max_iterations=1000, lr_min=2e-4 / 100)
        lrs = []
        for i in range(1000):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"lr/{name}"] = [lrs[i] for i in (0, 1, 9, 10, 99, 100, 101, 299, 300, 500, 699, 700, 999)]
    with open(os.path.join(HERE, "host_helpers.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("host_helpers:", sorted(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["census", "tiny_forward", "loss", "schema", "lora_init", "host_helpers", "lora_init_image"]
    for w in which:
        {"census": census, "tiny_forward": tiny_forward, "loss": loss, "schema": schema, "lora_init": lora_init,
         "host_helpers": host_helpers, "lora_init_image": lora_init_image}[w]()
