"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol declared in
include/rtdiff.h, and its weight table equals the reference UNet's state_dict layout (oracle/shapes.py,
itself pinned against the reference's own state_dict in test_oracle_vs_reference.py)."""
import os
import re

import pytest

from oracle.shapes import weight_shapes
from oracle.unet import SD15_CONFIG, SDXL_CONFIG, TINY_SD_CONFIG, TINY_XL_CONFIG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "rtdiff.h")).read()
    names = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    names -= {"rt_config", "rt_engine"}
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), f"librtdiff.so does not export {n}"


@pytest.mark.parametrize("cfg", [SD15_CONFIG, SDXL_CONFIG, TINY_SD_CONFIG, TINY_XL_CONFIG], ids=["sd15", "sdxl", "tiny_sd", "tiny_xl"])
def test_weight_table_matches_reference_state_dict_layout(cfg):
    from rich_text_to_image_amd.engine import Engine
    e = Engine(cfg, 64, 64, device=-1)          # weight-table-only engine: no GPU needed
    table = dict(e.weight_table())
    shapes = {k: tuple(v) for k, v in weight_shapes(cfg).items()}
    assert set(table) == set(shapes)
    for k, v in shapes.items():
        assert tuple(table[k]) == v, k
    n, missing = e.weights_missing()
    assert n == len(shapes)
    e.close()


def test_parameter_counts_match_published_checkpoints():
    def count(cfg):
        n = 0
        for v in weight_shapes(cfg).values():
            k = 1
            for d in v:
                k *= d
            n += k
        return n
    assert count(SD15_CONFIG) == 859_520_964          # runwayml/stable-diffusion-v1-5 unet
    assert count(SDXL_CONFIG) == 2_567_463_684        # stabilityai/stable-diffusion-xl-base-1.0 unet


def test_device_calls_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rich_text_to_image_amd.engine import Engine, RtError
    e = Engine(TINY_XL_CONFIG, 32, 32, device=-1)
    with pytest.raises(RtError):
        e.synchronize()
    with pytest.raises(RtError):
        e.region_step(0, 5.0)
    e.close()


def test_invalid_config_is_rejected():
    from rich_text_to_image_amd.engine import Engine, RtError
    bad = dict(TINY_XL_CONFIG)
    with pytest.raises(RtError):
        Engine(bad, 32, 32, device=-1, max_streams=99)
    bad = dict(TINY_XL_CONFIG, down_block_types=("AttnDownBlock2D",) * 3)
    with pytest.raises(ValueError):
        Engine(bad, 32, 32, device=-1)


def test_vae_weight_table_matches_autoencoderkl_decoder_layout():
    from oracle.vae import SD_VAE_CONFIG, TINY_VAE_CONFIG, vae_decoder_shapes
    from rich_text_to_image_amd.engine import VaeDecoder
    for cfg in (SD_VAE_CONFIG, TINY_VAE_CONFIG):
        v = VaeDecoder(cfg, 64, 64, device=-1)                 # weight-table-only (no GPU)
        assert {n: tuple(s) for n, s in v.weight_table()} == {k: tuple(s) for k, s in vae_decoder_shapes(cfg).items()}
        v.close()
