"""The oracle (oracle/unet_oracle.py: the nn.Module restatement of diffusers-0.20.2's UNet2DConditionModel that every epsilon in
this repo is compared with) against a SECOND restatement written independently from SURVEY.md Appendix A (oracle/unet_f64.py:
functional, no nn.Module, no torch.nn.functional convolution / normalisation / attention / GELU).  Both in float64: any
disagreement above rounding noise is a restatement error in one of them (block order, skip wiring, head split, which norm has
which eps, GEGLU halves, [cos | sin] order, SDXL's added conditioning).  SD-1.x (conv proj_in / out, 8 heads), SD-2.x (linear proj,
64-wide heads) and SDXL (no attention in the first block, text_time embedding) topologies at reduced width, the CFG pair with
two DIFFERENT samples so that batch mix-ups show.  CPU only."""
import pytest
import torch

from oracle.unet_f64 import FunctionalUNetF64
from oracle.unet_oracle import CONFIGS, build_unet


@pytest.mark.parametrize("name,t", [("tiny_sd1", 981), ("tiny_sd2", 400), ("tiny_sdxl", 19), ("tiny_sdxl", 999)])
def test_two_independent_restatements_agree_in_float64(name, t):
    cfg = CONFIGS[name]()
    net = build_unet(name, seed=5, dtype=torch.float64)
    # LayerNorm / GroupNorm affine parameters and biases away from their (1, 0) initial values: a swapped or dropped affine must show
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for n, prm in net.named_parameters():
            if prm.ndim == 1:
                prm.add_(0.3 * torch.randn(prm.shape, generator=g, dtype=torch.float64))
    hw = 16
    x = torch.randn(2, 4, hw, hw, generator=g, dtype=torch.float64)
    ctx = torch.randn(2, 77, cfg.cross_attention_dim, generator=g, dtype=torch.float64)
    kw = None
    if cfg.addition_embed_type:
        pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        kw = {"text_embeds": torch.randn(2, pooled, generator=g, dtype=torch.float64),
              "time_ids": torch.tensor([[128.0, 96.0, 8.0, 0.0, 128.0, 128.0], [64.0, 128.0, 0.0, 16.0, 96.0, 128.0]], dtype=torch.float64)}
    with torch.no_grad():
        ref = net(x, torch.tensor(t), ctx, kw).sample
        got = FunctionalUNetF64(cfg, net.state_dict())(x, t, ctx, kw)
    assert got.shape == ref.shape == (2, 4, hw, hw)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    print(f"[parity] oracle vs independent fp64 restatement, {name} t={t}: max abs diff {err:.3e} at |eps|max {scale:.3f}")
    assert scale > 1e-2 and torch.isfinite(got).all()
    assert err <= 1e-10 * max(1.0, scale), f"{name}: the two restatements differ by {err:.3e}"
    # the two samples of the pair are different predictions (a batch mix-up would not show otherwise)
    assert (ref[0] - ref[1]).abs().max().item() > 1e-3


def test_independent_restatement_covers_every_parameter():
    """Every tensor of the oracle's state dict is read by the functional restatement (a parameter it never touches would be a module
    one of the two derivations does not have), for all three full-size configurations - checked on the names alone."""
    for name in ("sd1", "sd2", "sdxl"):
        cfg = CONFIGS[name]()
        net = build_unet(name, device="meta")
        names = set(net.state_dict().keys())

        # names the functional forward would touch, derived by a dry walk over its own naming scheme
        touched = set()
        boc, L = cfg.block_out_channels, cfg.layers_per_block

        def lin(n, bias=True):
            touched.add(n + ".weight")
            if bias:
                touched.add(n + ".bias")

        def resnet(n, shortcut):
            for s in ("norm1", "conv1", "time_emb_proj", "norm2", "conv2"):
                lin(f"{n}.{s}")
            if shortcut:
                lin(n + ".conv_shortcut")

        def transformer(n, layers):
            for s in ("norm", "proj_in", "proj_out"):
                lin(f"{n}.{s}")
            for k in range(layers):
                b = f"{n}.transformer_blocks.{k}"
                for a in ("attn1", "attn2"):
                    for s in ("to_q", "to_k", "to_v"):
                        lin(f"{b}.{a}.{s}", bias=False)
                    lin(f"{b}.{a}.to_out.0")
                for s in ("norm1", "norm2", "norm3", "ff.net.0.proj", "ff.net.2"):
                    lin(f"{b}.{s}")
        for s in ("time_embedding.linear_1", "time_embedding.linear_2", "conv_in", "conv_norm_out", "conv_out"):
            lin(s)
        if cfg.addition_embed_type:
            lin("add_embedding.linear_1")
            lin("add_embedding.linear_2")
        cin = boc[0]
        skip_c = [boc[0]]
        for i, typ in enumerate(cfg.down_block_types):
            for j in range(L):
                resnet(f"down_blocks.{i}.resnets.{j}", cin != boc[i])
                cin = boc[i]
                if typ == "CrossAttnDownBlock2D":
                    transformer(f"down_blocks.{i}.attentions.{j}", cfg.transformer_layers_per_block[i])
                skip_c.append(cin)
            if i != len(boc) - 1:
                lin(f"down_blocks.{i}.downsamplers.0.conv")
                skip_c.append(cin)
        resnet("mid_block.resnets.0", False)
        transformer("mid_block.attentions.0", cfg.transformer_layers_per_block[-1])
        resnet("mid_block.resnets.1", False)
        rboc, rl = tuple(reversed(boc)), tuple(reversed(cfg.transformer_layers_per_block))
        for i, typ in enumerate(cfg.up_block_types):
            for j in range(L + 1):
                resnet(f"up_blocks.{i}.resnets.{j}", cin + skip_c.pop() != rboc[i])
                cin = rboc[i]
                if typ == "CrossAttnUpBlock2D":
                    transformer(f"up_blocks.{i}.attentions.{j}", rl[i])
            if i != len(boc) - 1:
                lin(f"up_blocks.{i}.upsamplers.0.conv")
        assert touched == names, (name, sorted(touched ^ names)[:10])

