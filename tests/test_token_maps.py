"""Token-map producer: the get_token_maps port against outputs of the REFERENCE function
(tests/golden/token_maps_port.pt, oracle/make_golden.py tokenmaps) on regenerated synthetic maps -- CPU."""
import os

import torch

from oracle.synth import synthetic_attention_maps

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_get_token_maps_port_matches_reference_function():
    from rich_text_to_image_amd.attention_utils import get_token_maps
    port = torch.load(os.path.join(GOLD, "token_maps_port.pt"))
    assert len(port) == 2
    for (seed, nseg, thr), ref in port.items():
        selfm, crossm = synthetic_attention_maps(seed)
        masks = get_token_maps(selfm, crossm, {}, None, 64, 64, [torch.tensor([2, 3]), torch.tensor([6])], seed=4,
                               segment_threshold=thr, num_segments=nseg, device="cpu")
        got = torch.cat(masks)[:, 0]
        assert got.shape == ref.shape
        assert len(masks) == 3 and masks[0].shape == (1, 4, 64, 64)
        assert torch.allclose(got, ref, atol=1e-6), (got - ref).abs().max()
        assert torch.allclose(torch.cat(masks).sum(0), torch.ones(4, 64, 64), atol=1e-5)      # regions partition the image
        # the synthetic blobs are recovered: span {2,3} covers blobs 0 and 1, span {6} covers blob 3
        assert got[0, 6 * 2, 6 * 2] > 0.9 and got[0, 6 * 2, 24 * 2] > 0.9 and got[1, 24 * 2, 25 * 2] > 0.9 and got[2, 30, 32] > 0.9


def test_layer_lists_match_the_reference_names():
    from rich_text_to_image_amd import attention_utils as au
    from rich_text_to_image_amd.engine import Engine, SD15_CONFIG, SDXL_CONFIG
    for cfg, lists in ((SD15_CONFIG, au.SelfAttentionLayers + au.CrossAttentionLayers), (SDXL_CONFIG, au.CrossAttentionLayers_XL)):
        e = Engine(cfg, 64, 64, device=-1)
        names = {n for n, _, _ in e.attn_modules()}
        assert set(lists) <= names
        e.close()
