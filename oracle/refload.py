"""TEST INFRASTRUCTURE ONLY: import the unmodified reference (/root/reference) in the build container.

Used by oracle/make_golden.py and tests/test_oracle_vs_reference.py (the latter is
skipped when /root/reference is absent, i.e. on the GPU box). Never imported by the
product package.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("RTDIFF_REFERENCE", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def load_reference():
    """Returns dict of reference modules (unet_2d_condition, attention_processor,
    region_diffusion, region_diffusion_sdxl)."""
    if not reference_available():
        raise RuntimeError("reference tree not present")
    import transformers  # noqa: F401  (must be imported before the fake torchvision appears)
    for name in ("seaborn",):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    for p in (_SHIM, REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    mods = {}
    for m in ("models.attention_processor", "models.unet_2d_condition",
              "models.region_diffusion", "models.region_diffusion_sdxl"):
        mods[m.split(".")[-1]] = importlib.import_module(m)
    return mods
