"""head-averaged attention maps recorded on the GPU during the plain pass, against the maps the
REFERENCE hooks recorded (tests/golden/tokenmaps_*.pt: rd.py:397-443 / xl.py:959-1016 driven by oracle/make_golden.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.unet import TINY_SD_CONFIG, TINY_XL_CONFIG, random_state_dict  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


def _check(model, g):
    for k in g["self_names"]:
        assert k in model.selfattn_maps, k
        m = model.selfattn_maps[k].cpu()
        assert m.shape == (1, 1024, 1024)
        r = rel_l2(m[0, ::16], g["self_maps_rows"][k])
        rs = (m[0].sum(-1) - g["self_maps_rowsum"][k]).abs().max().item()
        print(f"self {k}: rel-L2 {r:.3e}, rowsum err {rs:.2e}")
        assert r < 3e-2 and rs < 2e-2 * g["self_maps_rowsum"][k].max().item()
    for k in g["cross_names"]:
        m = model.crossattn_maps[k].cpu()
        r = rel_l2(m[0], g["cross_maps"][k])
        print(f"cross {k}: rel-L2 {r:.3e}")
        assert m.shape[-1] == 77 and r < 3e-2
    assert set(int(v) for v in model.n_maps.values()) == set(g["n_maps_values"])


def test_sd_plain_pass_attention_store_matches_reference_hooks():
    from rich_text_to_image_amd.attention_utils import get_token_maps
    from rich_text_to_image_amd.region_diffusion import RegionDiffusion
    g = torch.load(os.path.join(GOLD, "tokenmaps_sd.pt"))
    m = RegionDiffusion(0, unet_state_dict=random_state_dict(TINY_SD_CONFIG, seed=g["weight_seed"]), config=TINY_SD_CONFIG)
    m.register_tokenmap_hooks()
    lat = m.plain_latents(g["embeds"], num_inference_steps=g["steps"], guidance_scale=g["guidance_scale"], latents=g["latents"].clone())
    assert rel_l2(lat, g["final_latents"]) < 3e-2
    _check(m, g)
    masks = get_token_maps(m.selfattn_maps, m.crossattn_maps, m.n_maps, None, 64, 64, g["obj_tokens"], seed=3)
    assert len(masks) == 3 and masks[0].shape == (1, 4, 64, 64) and masks[0].is_cuda
    assert torch.allclose(torch.cat(masks).sum(0).cpu(), torch.ones(4, 64, 64), atol=1e-4)
    m.remove_tokenmap_hooks()
    assert m.selfattn_maps is None


def test_xl_plain_pass_attention_store_matches_reference_hooks():
    from rich_text_to_image_amd.region_diffusion_sdxl import RegionDiffusionXL
    g = torch.load(os.path.join(GOLD, "tokenmaps_xl.pt"))
    m = RegionDiffusionXL(device=0, unet_state_dict=random_state_dict(TINY_XL_CONFIG, seed=g["weight_seed"]), config=TINY_XL_CONFIG)
    m.cross_attention_layers = g["xl_cross_layers"]
    m.register_tokenmap_hooks()
    out = m.sample(prompt=None, height=1024, width=1024, num_inference_steps=g["steps"], guidance_scale=g["guidance_scale"],
                   latents=g["latents"].clone(), prompt_embeds=g["embeds"][1:], negative_prompt_embeds=g["embeds"][:1],
                   pooled_prompt_embeds=g["pooled"][1:], negative_pooled_prompt_embeds=g["pooled"][:1], output_type="latent",
                   run_rich_text=False, original_size=(1024, 1024), target_size=(1024, 1024)).images
    assert rel_l2(out, g["final_latents"]) < 3e-2
    _check(m, g)


def test_get_token_maps_on_gpu_resident_maps_matches_reference_function_golden():
    """The PRODUCTION hand-over: the facades give get_token_maps CUDA tensors, which are averaged on the GPU (another fp32 reduction
    order than the reference's CPU mean) before the seeded clustering on the host.  Same synthetic maps as tests/test_token_maps.py,
    as CUDA tensors, against the outputs of the REFERENCE function (tests/golden/token_maps_port.pt): the cluster labels must not flip."""
    import os
    from oracle.synth import synthetic_attention_maps
    from rich_text_to_image_amd.attention_utils import get_token_maps
    port = torch.load(os.path.join(os.path.dirname(__file__), "golden", "token_maps_port.pt"))
    for (seed, nseg, thr), ref in port.items():
        selfm, crossm = synthetic_attention_maps(seed)
        selfm = {k: v.to("cuda:0") for k, v in selfm.items()}
        crossm = {k: v.to("cuda:0") for k, v in crossm.items()}
        masks = get_token_maps(selfm, crossm, {}, None, 64, 64, [torch.tensor([2, 3]), torch.tensor([6])], seed=4,
                               segment_threshold=thr, num_segments=nseg, device="cuda:0")
        got = torch.cat(masks)[:, 0].cpu()
        assert masks[0].is_cuda and got.shape == ref.shape
        assert torch.allclose(got, ref, atol=1e-5), (got - ref).abs().max()
