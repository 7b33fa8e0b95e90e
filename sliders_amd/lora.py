"""LoRANetwork: host-side mirror of the reference's adapter container
(trainscripts/textsliders/lora.py:115-258) over the MI355X UNetEngine.

Same constructor, same methods, same checkpoint layout:

    network = LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    with network: ...                       # adapters on (multiplier = lora_scale); off (0) outside
    network.set_lora_slider(scale)
    network.prepare_optimizer_params()      # one flat bf16 Parameter instead of 692 tiny ones
    network.save_weights(path, dtype)       # keys / shapes of lora.py:231-248 -> loads in the reference notebooks
    network.load_state_dict(torch.load(path))

The adapter arithmetic itself runs inside the engine's command buffers (slh_skinny + the slh_gemm epilogue);
this class only owns the packed parameter store and flips the engine's adapter switch.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .lora_store import LoraStore
from .modules import TRAINING_METHODS  # noqa: F401  (re-exported like the reference module does)
from .unet import UNetEngine

LORA_PREFIX_UNET = "lora_unet"
UNET_TARGET_REPLACE_MODULE_TRANSFORMER = ["Attention"]
UNET_TARGET_REPLACE_MODULE_CONV = ["ResnetBlock2D", "Downsample2D", "Upsample2D", "DownBlock2D", "UpBlock2D"]
# the reference's train scripts extend this list in place to switch c3lier on (train_lora.py:44-46)
DEFAULT_TARGET_REPLACE = UNET_TARGET_REPLACE_MODULE_TRANSFORMER


class LoRANetwork:
    def __init__(self, unet: UNetEngine, rank: int = 4, multiplier: float = 1.0, alpha: float = 1.0,
                 train_method: str = "full", kaiming_a: float = 1.0) -> None:
        self.lora_scale = 1
        self.multiplier = multiplier
        self.lora_dim = rank
        self.alpha = alpha
        self.unet = unet
        conv = [c for c in UNET_TARGET_REPLACE_MODULE_CONV if c in DEFAULT_TARGET_REPLACE]
        # the image sliders' list has no DownBlock2D / UpBlock2D (imagesliders/lora.py:19-25): same leaves, one RNG draw each
        ntype = "lierla" if not conv else ("c3lier" if "DownBlock2D" in conv else "c3lier-image")
        self.store = LoraStore(unet.cfg, rank=rank, alpha=alpha, train_method=train_method, network_type=ntype,
                               device=unet.device, kaiming_a=kaiming_a)
        self.unet_loras = self.store.entries
        print(f"create LoRA for U-Net: {len(self.unet_loras)} modules.")
        unet.attach_lora(self.store)
        # quirk kept: adapters are live at `multiplier` until the first __exit__
        unet.set_lora(multiplier != 0, multiplier)
        self._flat_param: Optional[torch.nn.Parameter] = None

    # nn.Module-ish surface the reference scripts touch
    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag=True):
        return self

    def train(self, mode=True):
        return self

    def eval(self):
        return self

    def parameters(self):
        return [self.flat_parameter()]

    def flat_parameter(self) -> torch.nn.Parameter:
        """The whole adapter state as ONE bf16 Parameter (a view of the packed buffer) so any torch optimizer
        can drive it; .grad is filled (bf16, like the reference's param.grad) by UNetEngine's autograd bridge."""
        if self._flat_param is None:
            self._flat_param = torch.nn.Parameter(self.store.params, requires_grad=True)
            self.unet.autograd_param = self._flat_param
        return self._flat_param

    def prepare_optimizer_params(self):
        return [{"params": [self.flat_parameter()]}]

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict: bool = True):
        self.store.load_state_dict(sd, strict=strict)

    def save_weights(self, file, dtype=None, metadata: Optional[dict] = None):
        sd = self.store.state_dict(dtype)
        if os.path.splitext(str(file))[1] == ".safetensors":
            from safetensors.torch import save_file
            save_file({k: v.contiguous() for k, v in sd.items()}, str(file), metadata)
        else:
            torch.save(sd, file)

    def set_lora_slider(self, scale):
        self.lora_scale = scale

    def __enter__(self):
        self.unet.set_lora(True, 1.0 * self.lora_scale)

    def __exit__(self, exc_type, exc_value, tb):
        self.unet.set_lora(False)
