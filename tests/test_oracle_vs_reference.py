"""Build-container only (needs /root/reference): pins the oracle against the UNMODIFIED reference code.
Skipped on the GPU box, where the committed fixtures (tests/golden) carry the same evidence."""
import pytest
import torch

from oracle.refload import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not present")

from oracle.shapes import weight_shapes  # noqa: E402
from oracle.unet import (SD15_CONFIG, SDXL_CONFIG, TINY_SD_CONFIG, TINY_XL_CONFIG, OracleUNet, random_state_dict,  # noqa: E402
                         reference_kwargs)


@pytest.fixture(scope="module")
def ref():
    from oracle.refload import load_reference
    return load_reference()


@pytest.mark.parametrize("cfg,xl", [(TINY_XL_CONFIG, True), (TINY_SD_CONFIG, False)], ids=["tiny_xl", "tiny_sd"])
def test_state_dict_layout_and_forward(ref, cfg, xl):
    U = ref["unet_2d_condition"].UNet2DConditionModel
    m = U(**reference_kwargs(cfg))
    rsd = m.state_dict()
    shapes = weight_shapes(cfg)
    assert set(shapes) == set(rsd)
    assert all(tuple(rsd[k].shape) == tuple(v) for k, v in shapes.items())
    sd = random_state_dict(cfg, seed=1)
    m.load_state_dict(sd)
    torch.manual_seed(0)
    B = 2
    x, ctx = torch.randn(B, 4, 32, 32), torch.randn(B, 77, cfg["cross_attention_dim"])
    added = {"text_embeds": torch.randn(B, 32), "time_ids": torch.tensor([[32., 32, 0, 0, 32, 32]] * B)} if xl else None
    with torch.no_grad():
        r = m(x, torch.tensor(481), encoder_hidden_states=ctx, added_cond_kwargs=added)["sample"]
        y = OracleUNet(cfg, sd).forward(x, 481, ctx, added)
    assert (r - y).abs().max().item() < 1e-5


@pytest.mark.parametrize("cfg", [SD15_CONFIG, SDXL_CONFIG], ids=["sd15", "sdxl"])
def test_full_size_layout_on_meta_device(ref, cfg):
    U = ref["unet_2d_condition"].UNet2DConditionModel
    with torch.device("meta"):
        m = U(**reference_kwargs(cfg))
    rsd = m.state_dict()
    shapes = weight_shapes(cfg)
    assert set(shapes) == set(rsd)
    assert all(tuple(rsd[k].shape) == tuple(v) for k, v in shapes.items())
