"""Single-file (LDM key layout) checkpoint conversion: known key pairs of the published SD-1.x / SDXL layouts, and
a full round trip over every UNet key of all four configs (no tensor may be dropped, duplicated or renamed twice)."""
import pytest
import torch

from oracle.unet_oracle import build_unet
from sliders_amd.config import CONFIGS
from sliders_amd.ldm_convert import (LDM_PREFIX, convert_ldm_unet_state_dict, diffusers_to_ldm_key,
                                     ldm_to_diffusers_key, load_single_file_unet)

KNOWN_SD1 = {
    "time_embed.0.weight": "time_embedding.linear_1.weight",
    "time_embed.2.bias": "time_embedding.linear_2.bias",
    "input_blocks.0.0.weight": "conv_in.weight",
    "input_blocks.1.0.in_layers.0.weight": "down_blocks.0.resnets.0.norm1.weight",
    "input_blocks.1.0.emb_layers.1.bias": "down_blocks.0.resnets.0.time_emb_proj.bias",
    "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight": "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
    "input_blocks.3.0.op.weight": "down_blocks.0.downsamplers.0.conv.weight",
    "input_blocks.4.0.skip_connection.weight": "down_blocks.1.resnets.0.conv_shortcut.weight",
    "input_blocks.11.0.out_layers.3.weight": "down_blocks.3.resnets.1.conv2.weight",
    "middle_block.1.proj_in.weight": "mid_block.attentions.0.proj_in.weight",
    "middle_block.2.in_layers.2.bias": "mid_block.resnets.1.conv1.bias",
    "output_blocks.2.1.conv.weight": "up_blocks.0.upsamplers.0.conv.weight",
    "output_blocks.5.2.conv.weight": "up_blocks.1.upsamplers.0.conv.weight",
    "output_blocks.5.1.transformer_blocks.0.ff.net.0.proj.weight": "up_blocks.1.attentions.2.transformer_blocks.0.ff.net.0.proj.weight",
    "output_blocks.11.0.out_layers.0.weight": "up_blocks.3.resnets.2.norm2.weight",
    "out.0.weight": "conv_norm_out.weight",
    "out.2.bias": "conv_out.bias",
}
KNOWN_SDXL = {
    "label_emb.0.0.weight": "add_embedding.linear_1.weight",
    "label_emb.0.2.bias": "add_embedding.linear_2.bias",
    "input_blocks.2.0.in_layers.2.weight": "down_blocks.0.resnets.1.conv1.weight",
    "input_blocks.4.1.transformer_blocks.1.attn2.to_k.weight": "down_blocks.1.attentions.0.transformer_blocks.1.attn2.to_k.weight",
    "input_blocks.6.0.op.bias": "down_blocks.1.downsamplers.0.conv.bias",
    "input_blocks.8.1.transformer_blocks.9.ff.net.2.weight": "down_blocks.2.attentions.1.transformer_blocks.9.ff.net.2.weight",
    "output_blocks.2.2.conv.weight": "up_blocks.0.upsamplers.0.conv.weight",
    "output_blocks.5.2.conv.bias": "up_blocks.1.upsamplers.0.conv.bias",
    "output_blocks.8.0.skip_connection.weight": "up_blocks.2.resnets.2.conv_shortcut.weight",
}


@pytest.mark.parametrize("name,known", [("sd1", KNOWN_SD1), ("sdxl", KNOWN_SDXL)])
def test_known_key_pairs(name, known):
    cfg = CONFIGS[name]()
    for ldm, dif in known.items():
        assert ldm_to_diffusers_key(ldm, cfg) == dif
        assert diffusers_to_ldm_key(dif, cfg) == ldm


@pytest.mark.parametrize("name", ["sd1", "sdxl", "tiny_sd1", "tiny_sdxl"])
def test_round_trip_over_every_unet_key(name):
    cfg = CONFIGS[name]()
    keys = list(build_unet(name, device="meta").state_dict().keys())
    ldm = [diffusers_to_ldm_key(k, cfg) for k in keys]
    assert len(set(ldm)) == len(keys), "two diffusers keys map to one LDM key"
    assert [ldm_to_diffusers_key(k, cfg) for k in ldm] == keys


def test_single_file_checkpoint_to_engine_state_dict(tmp_path):
    from safetensors.torch import save_file
    cfg = CONFIGS["tiny_sdxl"]()
    sd = {k: v.contiguous() for k, v in build_unet("tiny_sdxl", seed=1).state_dict().items()}
    single = {LDM_PREFIX + diffusers_to_ldm_key(k, cfg): v for k, v in sd.items()}
    single["first_stage_model.encoder.conv_in.weight"] = torch.zeros(4)          # VAE / text-encoder tensors are skipped
    single["conditioner.embedders.0.transformer.text_model.final_layer_norm.bias"] = torch.zeros(4)
    path = tmp_path / "sd_xl_tiny.safetensors"
    save_file(single, str(path))
    got = load_single_file_unet(str(path), cfg)
    assert got.keys() == sd.keys() and all(torch.equal(got[k], sd[k]) for k in sd)
    ck = tmp_path / "tiny.ckpt"
    torch.save({"state_dict": single}, ck)
    got = load_single_file_unet(str(ck), cfg)
    assert got.keys() == sd.keys()
    with pytest.raises(ValueError):
        convert_ldm_unet_state_dict({"foo.weight": torch.zeros(1)}, cfg)
    with pytest.raises(KeyError):
        convert_ldm_unet_state_dict({LDM_PREFIX + "bogus.0.weight": torch.zeros(1)}, cfg)


def test_model_util_detects_single_file_layout(tmp_path, monkeypatch):
    """load_unet_state on a file path: SDXL is recognised by its label_emb MLP, SD-2.x by its 2-D proj_in weight."""
    from safetensors.torch import save_file
    from sliders_amd import model_util
    cfg = CONFIGS["tiny_sdxl"]()
    monkeypatch.setitem(model_util.CONFIGS, "sdxl", CONFIGS["tiny_sdxl"])       # tiny stand-in with the SDXL topology
    sd = {k: v.contiguous() for k, v in build_unet("tiny_sdxl", seed=2).state_dict().items()}
    path = tmp_path / "xl.safetensors"
    save_file({LDM_PREFIX + diffusers_to_ldm_key(k, cfg): v for k, v in sd.items()}, str(path))
    got_cfg, got = model_util.load_unet_state(str(path))
    assert got_cfg == cfg and got.keys() == sd.keys()
    cfg2 = CONFIGS["tiny_sd2"]()
    monkeypatch.setitem(model_util.CONFIGS, "sd2", CONFIGS["tiny_sd2"])
    sd2 = {k: v.contiguous() for k, v in build_unet("tiny_sd2", seed=3).state_dict().items()}
    v2 = tmp_path / "v2.safetensors"
    save_file({LDM_PREFIX + diffusers_to_ldm_key(k, cfg2): v for k, v in sd2.items()}, str(v2))
    got_cfg, got = model_util.load_unet_state(str(v2))
    assert got_cfg == cfg2 and got.keys() == sd2.keys() and all(torch.equal(got[k], sd2[k]) for k in sd2)
