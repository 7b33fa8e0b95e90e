"""Import path the reference's inference notebooks use (`from trainscripts.textsliders.lora import LoRANetwork`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sliders_amd.lora import (DEFAULT_TARGET_REPLACE, LORA_PREFIX_UNET, LoRANetwork,  # noqa: E402,F401
                              UNET_TARGET_REPLACE_MODULE_CONV, UNET_TARGET_REPLACE_MODULE_TRANSFORMER)
