"""Abstract module tree of the diffusers-0.20.2 UNet2DConditionModel, derived from a UNetConfig.

The reference discovers its LoRA targets by walking `unet.named_modules()` and matching
`module.__class__.__name__` strings (trainscripts/textsliders/lora.py:164-218).  The MI355X engine has no
torch module tree, so this file rebuilds the same tree shape (names, class names, registration order,
leaf Linear/Conv2d geometry) from the config; `lora_targets()` then restates the reference's selection
logic over it.  tests/test_lora_census.py checks the result against golden key lists produced by the
reference's own LoRANetwork (tests/golden/make_golden.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Tuple

from .config import UNetConfig


@dataclass
class Node:
    name: str                      # local name ("resnets", "0", "conv1", ...)
    cls: str                       # diffusers class name
    children: List["Node"] = field(default_factory=list)
    # leaf geometry (Linear: kernel 0; Conv: kernel 1|3)
    in_dim: int = 0
    out_dim: int = 0
    kernel: int = 0
    stride: int = 1
    bias: bool = True

    def add(self, child: "Node") -> "Node":
        self.children.append(child)
        return child

    def named_modules(self, prefix: str = "") -> Iterator[Tuple[str, "Node"]]:
        """Pre-order walk with torch's dotted-name convention (root is '')."""
        yield prefix, self
        for c in self.children:
            sub = c.name if prefix == "" else prefix + "." + c.name
            yield from c.named_modules(sub)


def _linear(name, i, o, cls="Linear", bias=True):
    return Node(name, cls, in_dim=i, out_dim=o, kernel=0, bias=bias)


def _conv(name, i, o, k, stride=1, cls="Conv2d"):
    return Node(name, cls, in_dim=i, out_dim=o, kernel=k, stride=stride)


def _module_list(name, items):
    n = Node(name, "ModuleList")
    for i, it in enumerate(items):
        it.name = str(i)
        n.add(it)
    return n


def _resnet(cin, cout, temb):
    n = Node("", "ResnetBlock2D", in_dim=cin, out_dim=cout)
    n.add(Node("norm1", "GroupNorm", in_dim=cin))
    n.add(_conv("conv1", cin, cout, 3, cls="LoRACompatibleConv"))
    n.add(_linear("time_emb_proj", temb, cout, cls="LoRACompatibleLinear"))
    n.add(Node("norm2", "GroupNorm", in_dim=cout))
    n.add(Node("dropout", "Dropout"))
    n.add(_conv("conv2", cout, cout, 3, cls="LoRACompatibleConv"))
    n.add(Node("nonlinearity", "SiLU"))
    if cin != cout:
        n.add(_conv("conv_shortcut", cin, cout, 1, cls="LoRACompatibleConv"))
    return n


def _attention(name, qdim, kvdim, inner):
    n = Node(name, "Attention", in_dim=qdim, out_dim=inner)
    n.add(_linear("to_q", qdim, inner, bias=False))
    n.add(_linear("to_k", kvdim, inner, bias=False))
    n.add(_linear("to_v", kvdim, inner, bias=False))
    n.add(_module_list("to_out", [_linear("", inner, qdim), Node("", "Dropout")]))
    return n


def _tblock(dim, ctx_dim):
    n = Node("", "BasicTransformerBlock", in_dim=dim)
    n.add(Node("norm1", "LayerNorm", in_dim=dim))
    n.add(_attention("attn1", dim, dim, dim))
    n.add(Node("norm2", "LayerNorm", in_dim=dim))
    n.add(_attention("attn2", dim, ctx_dim, dim))
    n.add(Node("norm3", "LayerNorm", in_dim=dim))
    ff = n.add(Node("ff", "FeedForward"))
    geglu = Node("", "GEGLU")
    geglu.add(_linear("proj", dim, dim * 8, cls="LoRACompatibleLinear"))
    ff.add(_module_list("net", [geglu, Node("", "Dropout"), _linear("", dim * 4, dim, cls="LoRACompatibleLinear")]))
    return n


def _transformer(ch, layers, ctx_dim, linear_proj):
    n = Node("", "Transformer2DModel", in_dim=ch)
    n.add(Node("norm", "GroupNorm", in_dim=ch))
    if linear_proj:
        n.add(_linear("proj_in", ch, ch, cls="LoRACompatibleLinear"))
    else:
        n.add(_conv("proj_in", ch, ch, 1, cls="LoRACompatibleConv"))
    n.add(_module_list("transformer_blocks", [_tblock(ch, ctx_dim) for _ in range(layers)]))
    if linear_proj:
        n.add(_linear("proj_out", ch, ch, cls="LoRACompatibleLinear"))
    else:
        n.add(_conv("proj_out", ch, ch, 1, cls="LoRACompatibleConv"))
    return n


def _downsample(ch):
    n = Node("", "Downsample2D")
    n.add(_conv("conv", ch, ch, 3, stride=2, cls="LoRACompatibleConv"))
    return n


def _upsample(ch):
    n = Node("", "Upsample2D")
    n.add(_conv("conv", ch, ch, 3, cls="LoRACompatibleConv"))
    return n


def build_tree(cfg: UNetConfig) -> Node:
    boc = cfg.block_out_channels
    ted = cfg.time_embed_dim
    root = Node("", "UNet2DConditionModel")
    root.add(_conv("conv_in", cfg.in_channels, boc[0], 3))
    root.add(Node("time_proj", "Timesteps"))
    te = root.add(Node("time_embedding", "TimestepEmbedding"))
    te.add(_linear("linear_1", boc[0], ted))
    te.add(Node("act", "SiLU"))
    te.add(_linear("linear_2", ted, ted))
    if cfg.is_xl:
        root.add(Node("add_time_proj", "Timesteps"))
        ae = root.add(Node("add_embedding", "TimestepEmbedding"))
        ae.add(_linear("linear_1", cfg.projection_class_embeddings_input_dim, ted))
        ae.add(Node("act", "SiLU"))
        ae.add(_linear("linear_2", ted, ted))

    down = []
    out_ch = boc[0]
    for i, t in enumerate(cfg.down_block_types):
        in_ch, out_ch = out_ch, boc[i]
        final = i == len(boc) - 1
        resnets = [_resnet(in_ch if j == 0 else out_ch, out_ch, ted) for j in range(cfg.layers_per_block)]
        if t == "DownBlock2D":
            blk = Node("", "DownBlock2D")
            blk.add(_module_list("resnets", resnets))
        else:
            blk = Node("", "CrossAttnDownBlock2D")
            blk.add(_module_list("attentions", [
                _transformer(out_ch, cfg.transformer_layers_per_block[i], cfg.cross_attention_dim,
                             cfg.use_linear_projection) for _ in range(cfg.layers_per_block)]))
            blk.add(_module_list("resnets", resnets))
        if not final:
            blk.add(_module_list("downsamplers", [_downsample(out_ch)]))
        down.append(blk)
    root.add(_module_list("down_blocks", down))

    up = []
    rboc = tuple(reversed(boc))
    rtl = tuple(reversed(cfg.transformer_layers_per_block))
    out_ch = rboc[0]
    nl = cfg.layers_per_block + 1
    for i, t in enumerate(cfg.up_block_types):
        final = i == len(boc) - 1
        prev, out_ch = out_ch, rboc[i]
        in_ch = rboc[min(i + 1, len(boc) - 1)]
        resnets = []
        for j in range(nl):
            skip = in_ch if j == nl - 1 else out_ch
            rin = prev if j == 0 else out_ch
            resnets.append(_resnet(rin + skip, out_ch, ted))
        if t == "UpBlock2D":
            blk = Node("", "UpBlock2D")
            blk.add(_module_list("resnets", resnets))
        else:
            blk = Node("", "CrossAttnUpBlock2D")
            blk.add(_module_list("attentions", [
                _transformer(out_ch, rtl[i], cfg.cross_attention_dim, cfg.use_linear_projection)
                for _ in range(nl)]))
            blk.add(_module_list("resnets", resnets))
        if not final:
            blk.add(_module_list("upsamplers", [_upsample(out_ch)]))
        up.append(blk)
    root.add(_module_list("up_blocks", up))

    mid = root.add(Node("mid_block", "UNetMidBlock2DCrossAttn"))
    mid.add(_module_list("attentions", [
        _transformer(boc[-1], cfg.transformer_layers_per_block[-1], cfg.cross_attention_dim,
                     cfg.use_linear_projection)]))
    mid.add(_module_list("resnets", [_resnet(boc[-1], boc[-1], ted), _resnet(boc[-1], boc[-1], ted)]))

    root.add(Node("conv_norm_out", "GroupNorm", in_dim=boc[0]))
    root.add(Node("conv_act", "SiLU"))
    root.add(_conv("conv_out", boc[0], cfg.out_channels, 3))
    return root


# ---- the reference's target selection, restated (lora.py:15-30, 164-218) -----------------------------
UNET_TARGET_REPLACE_MODULE_TRANSFORMER = ["Attention"]
UNET_TARGET_REPLACE_MODULE_CONV = ["ResnetBlock2D", "Downsample2D", "Upsample2D", "DownBlock2D", "UpBlock2D"]
# trainscripts/imagesliders/lora.py:19-25 comments the two block classes out: the same leaves are reached (every conv under
# a DownBlock2D / UpBlock2D also sits under a ResnetBlock2D / Downsample2D / Upsample2D), but none twice - so the image
# sliders' network construction draws the RNG once per leaf, not once per visit (network_type "c3lier-image")
UNET_TARGET_REPLACE_MODULE_CONV_IMAGE = ["ResnetBlock2D", "Downsample2D", "Upsample2D"]
LORA_PREFIX_UNET = "lora_unet"
TRAINING_METHODS = ("noxattn", "innoxattn", "selfattn", "xattn", "full", "xattn-strict", "noxattn-hspace",
                    "noxattn-hspace-last")
_LEAF_CLASSES = ("Linear", "Conv2d", "LoRACompatibleLinear", "LoRACompatibleConv")


@dataclass
class LoraTarget:
    lora_name: str      # e.g. lora_unet_down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q
    module_path: str    # dotted diffusers path of the wrapped Linear / Conv2d
    kind: str           # "linear" | "conv3" | "conv1"
    in_dim: int
    out_dim: int
    stride: int
    rank: int


def lora_targets(cfg: UNetConfig, train_method: str, rank: int = 4, network_type: str = "c3lier") -> List[LoraTarget]:
    """network_type 'c3lier' adds the conv classes (train_lora.py:44-46 mutates the shared list)."""
    return [t for t, dup in lora_visits(cfg, train_method, rank, network_type) if not dup]


def lora_visits(cfg: UNetConfig, train_method: str, rank: int = 4, network_type: str = "c3lier"):
    """Every (target, is_duplicate) the reference's create_modules CONSTRUCTS, in its order (lora.py:164-218): with the
    conv classes enabled, DownBlock2D / UpBlock2D are targets too, so every leaf under them is reached a second time
    through its ResnetBlock2D / Downsample2D / Upsample2D parent; the reference builds a LoRAModule for that second
    visit (drawing the RNG) and only then drops it by name.  Duplicates matter for seed parity of the initial weights."""
    if train_method not in TRAINING_METHODS:
        raise NotImplementedError(f"train_method: {train_method} is not implemented.")
    targets_cls = list(UNET_TARGET_REPLACE_MODULE_TRANSFORMER)
    if network_type == "c3lier":
        targets_cls += UNET_TARGET_REPLACE_MODULE_CONV
    elif network_type == "c3lier-image":
        targets_cls += UNET_TARGET_REPLACE_MODULE_CONV_IMAGE
    elif network_type != "lierla":
        raise ValueError(f"network type {network_type!r}: lierla, c3lier (text sliders) or c3lier-image (image sliders)")
    root = build_tree(cfg)
    out, names = [], set()
    for name, module in root.named_modules():
        if train_method in ("noxattn", "noxattn-hspace", "noxattn-hspace-last"):
            if "attn2" in name or "time_embed" in name:
                continue
        elif train_method == "innoxattn":
            if "attn2" in name:
                continue
        elif train_method == "selfattn":
            if "attn1" not in name:
                continue
        elif train_method in ("xattn", "xattn-strict"):
            if "attn2" not in name:
                continue
        if module.cls not in targets_cls:
            continue
        for child_name, child in module.named_modules():
            if child.cls not in _LEAF_CLASSES:
                continue
            if train_method == "xattn-strict" and "out" in child_name:
                continue
            if train_method == "noxattn-hspace" and "mid_block" not in name:
                continue
            if train_method == "noxattn-hspace-last":
                if "mid_block" not in name or ".1" not in name or "conv2" not in child_name:
                    continue
            lora_name = (LORA_PREFIX_UNET + "." + name + "." + child_name).replace(".", "_")
            dup = lora_name in names
            names.add(lora_name)
            if child.kernel == 0:
                kind, r = "linear", rank
            else:
                kind = "conv3" if child.kernel == 3 else "conv1"
                r = min(rank, child.in_dim, child.out_dim)   # lora.py:78
            out.append((LoraTarget(lora_name, name + "." + child_name, kind, child.in_dim, child.out_dim,
                                   child.stride, r), dup))
    return out
