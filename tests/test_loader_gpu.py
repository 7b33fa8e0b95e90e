"""Checkpoint files -> engine on the GPU: the two on-disk layouts the reference loads through
diffusers (model_util.py:67-72 directory with unet/config.json + safetensors; model_util.py:77-101 single-file LDM layout)
must give the SAME engine output as handing the engine the state dict directly, and that output must match the oracle
built from the same tensors.  No real checkpoint exists offline: the files are written from seeded random weights."""
import json

import pytest
import torch

from oracle.unet_oracle import build_unet
from sliders_amd.config import CONFIGS
from sliders_amd.ldm_convert import LDM_PREFIX, diffusers_to_ldm_key
from sliders_amd import model_util
from sliders_amd.unet import UNetEngine
from tests.test_unet_gpu import make_inputs, run_engine, run_oracle
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,stand_in", [("tiny_sdxl", "sdxl"), ("tiny_sd2", "sd2"), ("tiny_sd1", "sd1")])
def test_checkpoint_files_load_into_the_engine(dev, tmp_path, monkeypatch, name, stand_in):
    from safetensors.torch import save_file
    cfg = CONFIGS[name]()
    net = build_unet(name, seed=4)
    sd = {k: v.contiguous() for k, v in net.state_dict().items()}
    # diffusers directory
    unet_dir = tmp_path / "model" / "unet"
    unet_dir.mkdir(parents=True)
    raw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()}
    raw.update({"_class_name": "UNet2DConditionModel", "_diffusers_version": "0.20.2"})
    (unet_dir / "config.json").write_text(json.dumps(raw))
    save_file(sd, str(unet_dir / "diffusion_pytorch_model.safetensors"))
    # single-file LDM layout (the tiny config stands in for the full one the loader would pick from the key pattern)
    monkeypatch.setitem(model_util.CONFIGS, stand_in, CONFIGS[name])
    single = tmp_path / "single.safetensors"
    save_file({LDM_PREFIX + diffusers_to_ldm_key(k, cfg): v for k, v in sd.items()}, str(single))

    x, ctx, kw = make_inputs(cfg, 2, 16)
    e32 = run_oracle(net, x, 500, ctx, kw, torch.float32)
    outs = {}
    for tag, eng in (("state dict", UNetEngine(cfg, sd, dev)),
                     ("diffusers directory", model_util.load_unet_engine(str(tmp_path / "model"), dev)),
                     ("single file", model_util.load_unet_engine(str(single), dev))):
        assert eng.cfg == cfg
        outs[tag] = run_engine(eng, x, 500, ctx, kw, dev)
        r = rel_err(outs[tag], e32)
        print(f"[parity] {name} loaded from {tag}: rel_l2 vs fp32 oracle {r:.3e}")
        assert r < 2.0e-2
    # same weights, same plan: the three engines differ only by the run-to-run floor of the pass (GroupNorm atomics)
    for tag in ("diffusers directory", "single file"):
        assert rel_err(outs[tag], outs["state dict"]) < 1.5e-2
