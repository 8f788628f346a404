"""rtdiff-mi355x: MI355X-native region-diffusion sampling engine (drop-in for the sampling core of
songweige/rich-text-to-image).  Python here is plumbing only: tensors live in torch (ROCm), every
arithmetic op of the denoising hot path runs in librtdiff.so (hand-written HIP for gfx950)."""
from .engine import Engine, RtError, load_library, config_from_dict, SD15_CONFIG, SDXL_CONFIG  # noqa: F401

__all__ = ["Engine", "RtError", "load_library", "config_from_dict", "SD15_CONFIG", "SDXL_CONFIG"]
