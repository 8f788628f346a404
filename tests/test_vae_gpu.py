"""VAE decoder forward and the colour-guidance input gradient on the GPU against the oracle
restatement (oracle/vae.py; torch autograd supplies the reference gradient).  AutoencoderKL is third-party code that
is not on disk, so this row is pinned against the restatement only ("parity unpinned" vs diffusers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.vae import TINY_VAE_CONFIG, OracleVAEDecoder, color_guidance_update, random_vae_state_dict, vae_decoder_shapes  # noqa: E402

DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


@pytest.fixture(scope="module")
def vae():
    from rich_text_to_image_amd.engine import VaeDecoder
    sd = random_vae_state_dict(TINY_VAE_CONFIG, seed=1)
    v = VaeDecoder(TINY_VAE_CONFIG, 32, 32, device=0, state_dict=sd)
    assert {n: tuple(s) for n, s in v.weight_table()} == {k: tuple(s) for k, s in vae_decoder_shapes(TINY_VAE_CONFIG).items()}
    return v, OracleVAEDecoder(TINY_VAE_CONFIG, sd)


@pytest.mark.parametrize("hw", [16, 32])
def test_decode_matches_oracle(vae, hw):
    v, o = vae
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 4, hw, hw, generator=g) * 3
    with torch.no_grad():
        ref = o.decode(z)
    out = v.decode(z.to(DEV))
    r = rel_l2(out, ref)
    print(f"vae decode {hw}: rel-L2 {r:.3e} (ref rms {ref.pow(2).mean().sqrt():.3f})")
    assert out.shape == (1, 3, 8 * hw, 8 * hw) and r < 2e-2


def test_color_guidance_update_matches_autograd(vae):
    v, o = vae
    hw = 16
    g = torch.Generator().manual_seed(1)
    lat, eps = torch.randn(1, 4, hw, hw, generator=g), torch.randn(1, 4, hw, hw, generator=g)
    masks = [torch.rand(1, 4, 8 * hw, 8 * hw, generator=g) ** 2 for _ in range(2)]
    masks = [m[:, :1].repeat(1, 4, 1, 1) for m in masks]
    rgb = [torch.rand(1, 3, 1, 1, generator=g) for _ in range(2)]
    mall = torch.rand(1, 4, hw, hw, generator=g)
    alpha, sc, wgt = 0.37, TINY_VAE_CONFIG["scaling_factor"], 0.8
    new_ref, grad_ref, loss_ref = color_guidance_update(o, lat, eps, alpha, sc, masks, rgb, wgt, mall)
    lat_g = lat.clone().to(DEV)
    loss, grad = v.color_guidance(lat_g, eps.to(DEV), alpha, hw, hw, masks, rgb, wgt, mall, want_grad=True)
    rg = rel_l2(grad, grad_ref)
    ru = rel_l2(lat_g.cpu() - lat, new_ref - lat)
    print(f"colour guidance: loss {loss:.4f} vs {loss_ref:.4f}; grad rel-L2 {rg:.3e}; update rel-L2 {ru:.3e}; |grad|max {grad_ref.abs().max():.3e}")
    assert abs(loss - loss_ref) < 2e-2 * abs(loss_ref)
    assert rg < 5e-2 and ru < 5e-2


# ------------------------------------------------------------------------------------------------ precise mode (SDXL: fp32 VAE)
# The SDXL pipeline of the reference runs this decoder in fp32 (region_diffusion_sdxl.py:856).  precise=True builds fp32-class
# products from three bf16 MFMA passes over (hi, lo) operand pairs: 16 mantissa bits per operand, fp32 accumulation, fp32
# everywhere else.  Tolerances: decode 3e-4, gradient / update 1e-3 relative L2 against the fp32 oracle - two orders of magnitude
# inside the single-pass tolerances above (2e-2 / 5e-2); what remains is the dropped lo*lo term and fp32 summation order.
@pytest.fixture(scope="module")
def vae_precise():
    from rich_text_to_image_amd.engine import VaeDecoder
    sd = random_vae_state_dict(TINY_VAE_CONFIG, seed=1)
    return VaeDecoder(TINY_VAE_CONFIG, 32, 32, device=0, state_dict=sd, precise=True), OracleVAEDecoder(TINY_VAE_CONFIG, sd)


@pytest.mark.parametrize("hw", [16, 32])
def test_precise_decode_is_fp32_class(vae_precise, hw):
    v, o = vae_precise
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 4, hw, hw, generator=g) * 3
    with torch.no_grad():
        ref = o.decode(z)
    r = rel_l2(v.decode(z.to(DEV)), ref)
    print(f"precise vae decode {hw}: rel-L2 {r:.3e}")
    assert r < 3e-4


def test_precise_color_guidance_is_fp32_class(vae_precise):
    v, o = vae_precise
    hw = 16
    g = torch.Generator().manual_seed(1)
    lat, eps = torch.randn(1, 4, hw, hw, generator=g), torch.randn(1, 4, hw, hw, generator=g)
    masks = [(torch.rand(1, 1, 8 * hw, 8 * hw, generator=g) ** 2).repeat(1, 4, 1, 1) for _ in range(2)]
    rgb = [torch.rand(1, 3, 1, 1, generator=g) for _ in range(2)]
    mall = torch.rand(1, 4, hw, hw, generator=g)
    alpha, sc, wgt = 0.37, TINY_VAE_CONFIG["scaling_factor"], 0.8
    new_ref, grad_ref, loss_ref = color_guidance_update(o, lat, eps, alpha, sc, masks, rgb, wgt, mall)
    lat_g = lat.clone().to(DEV)
    loss, grad = v.color_guidance(lat_g, eps.to(DEV), alpha, hw, hw, masks, rgb, wgt, mall, want_grad=True)
    rg, ru = rel_l2(grad, grad_ref), rel_l2(lat_g.cpu() - lat, new_ref - lat)
    print(f"precise colour guidance: loss {loss:.6f} vs {loss_ref:.6f}; grad rel-L2 {rg:.3e}; update rel-L2 {ru:.3e}")
    assert abs(loss - loss_ref) < 1e-4 * abs(loss_ref)
    assert rg < 1e-3 and ru < 1e-3
