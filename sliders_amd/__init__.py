"""sliders_amd - MI355X-native concept-slider LoRA training hot path (UNet denoise step).

Host side mirrors the reference's Python interface (trainscripts/textsliders/*.py of
rohitgandikota/sliders); the arithmetic lives in libsliders_hip.so (hand-written HIP for gfx950,
C ABI in include/sliders_hip.h).  There is no CPU or PyTorch-op fallback for the hot path.
"""
__version__ = "0.1.0"
