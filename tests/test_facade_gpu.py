"""The drop-in classes reproduce the golden outputs of the reference classes they replace: same call, same
arguments (embeds / masks / latents / text_format_dict), outputs of the UNMODIFIED reference in tests/golden."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.unet import TINY_SD_CONFIG, TINY_XL_CONFIG, random_state_dict  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


def test_region_diffusion_produce_latents_matches_reference_class():
    from rich_text_to_image_amd.region_diffusion import RegionDiffusion
    g = torch.load(os.path.join(GOLD, "tiny_sd_plms.pt"))
    inp = g["inputs"]
    m = RegionDiffusion(0, unet_state_dict=random_state_dict(TINY_SD_CONFIG, seed=g["weight_seed"]), config=TINY_SD_CONFIG)
    m.masks = [x[None].repeat(1, 4, 1, 1) for x in inp["masks"]]
    out = m.produce_latents(inp["embeds"], height=512, width=512, num_inference_steps=g["steps"],
                            guidance_scale=g["guidance_scale"], latents=inp["latents"].clone(),
                            text_format_dict={"word_pos": inp["word_pos"], "font_size": inp["font_size"]},
                            inject_selfattn=g["inject_selfattn"], inject_background=g["inject_background"])
    r = rel_l2(out, g["reference_final_latents"])
    print("RegionDiffusion.produce_latents vs reference rel-L2", r)
    assert r < 3e-2
    m.masks = m.masks[:-1]
    with pytest.raises(AssertionError):                       # rd.py:97
        m.produce_latents(inp["embeds"], latents=inp["latents"].clone(), num_inference_steps=2)
    m.masks = [x[None].repeat(1, 4, 1, 1) for x in inp["masks"]]
    with pytest.raises(RuntimeError):                         # guidance needs the VAE decoder engine
        m.produce_latents(inp["embeds"], latents=inp["latents"].clone(), use_guidance=True, num_inference_steps=2)


def test_region_diffusion_xl_sample_matches_reference_class():
    from rich_text_to_image_amd.region_diffusion_sdxl import RegionDiffusionXL
    g = torch.load(os.path.join(GOLD, "tiny_xl_euler.pt"))
    inp = g["inputs"]
    m = RegionDiffusionXL(device=0, unet_state_dict=random_state_dict(TINY_XL_CONFIG, seed=g["weight_seed"]), config=TINY_XL_CONFIG)
    m.masks = [x[None].repeat(1, 4, 1, 1) for x in inp["masks"]]
    hw = inp["latents"].shape[2] * 8
    out = m.sample(prompt=None, height=hw, width=hw, num_inference_steps=g["steps"], guidance_scale=g["guidance_scale"],
                   latents=inp["latents"].clone(), prompt_embeds=inp["embeds"][1:], negative_prompt_embeds=inp["embeds"][:1],
                   pooled_prompt_embeds=inp["pooled"][1:], negative_pooled_prompt_embeds=inp["pooled"][:1],
                   output_type="latent", run_rich_text=True,
                   text_format_dict={"word_pos": inp["word_pos"], "font_size": inp["font_size"]},
                   inject_selfattn=g["inject_selfattn"], inject_background=g["inject_background"],
                   original_size=(hw, hw), target_size=(hw, hw)).images
    r = rel_l2(out, g["reference_final_latents"])
    print("RegionDiffusionXL.sample vs reference rel-L2", r)
    assert r < 3e-2
    with pytest.raises(ValueError):                           # check_inputs xl.py:462
        m.sample(prompt=None, height=100, width=hw, prompt_embeds=inp["embeds"][1:], pooled_prompt_embeds=inp["pooled"][1:])
    with pytest.raises(NotImplementedError):                  # xl.py:827-830
        m.sample(prompt=None, height=hw, width=hw, prompt_embeds=inp["embeds"][1:], negative_prompt_embeds=inp["embeds"][:1],
                 pooled_prompt_embeds=inp["pooled"][1:], negative_pooled_prompt_embeds=inp["pooled"][:1],
                 run_rich_text=True, guidance_rescale=0.7)


def test_unet_seam_call_signature():
    from rich_text_to_image_amd.unet import HipUNet2DConditionModel
    g = torch.load(os.path.join(GOLD, "tiny_sd_plms.pt"))
    u = HipUNet2DConditionModel(TINY_SD_CONFIG, random_state_dict(TINY_SD_CONFIG, seed=g["weight_seed"]))
    x = g["inputs"]["latents"].cuda()
    out = u(x, torch.tensor(481), encoder_hidden_states=g["inputs"]["embeds"][1:2].cuda())["sample"]
    assert rel_l2(out, g["reference_unet_t481"]) < 1.5e-2


@pytest.mark.parametrize("xl", [False, True], ids=["sd", "xl"])
def test_colour_guided_loop_matches_oracle_loop(xl):
    """Config-2 / config-5 style run: region loop + colour guidance (+ background blend) vs the oracle loop with the
    oracle VAE and torch autograd (rd.py:151-173 / xl.py:849-872)."""
    from oracle import region_loop
    from oracle.schedulers import OracleEuler, OraclePNDM
    from oracle.unet import OracleUNet
    from oracle.vae import TINY_VAE_CONFIG, OracleVAEDecoder, random_vae_state_dict
    from rich_text_to_image_amd.engine import VaeDecoder
    g = torch.load(os.path.join(GOLD, "tiny_xl_euler.pt" if xl else "tiny_sd_plms.pt"))
    inp = g["inputs"]
    cfg = TINY_XL_CONFIG if xl else TINY_SD_CONFIG
    hw = 128 if xl else 64                                     # the hook asserts fix the latent size (rd.py:339, xl.py:1091)
    sd = random_state_dict(cfg, seed=g["weight_seed"])
    vsd = random_vae_state_dict(TINY_VAE_CONFIG, seed=2)
    gen = torch.Generator().manual_seed(7)
    R = g["R"]
    lat = torch.randn(1, 4, hw, hw, generator=gen)
    m = torch.softmax(torch.randn(R, 1, hw, hw, generator=gen) * 2, 0).repeat(1, 4, 1, 1)
    masks = [m[r:r + 1] for r in range(R)]
    cm = [torch.rand(1, 1, 8 * hw, 8 * hw, generator=gen).repeat(1, 4, 1, 1) for _ in range(2)]
    tfd = {"word_pos": inp["word_pos"], "font_size": inp["font_size"], "target_RGB": [torch.rand(1, 3, 1, 1, generator=gen) for _ in range(2)],
           "guidance_start_step": 999, "color_guidance_weight": 0.5, "color_obj_atten": cm,
           "color_obj_atten_all": torch.rand(1, 4, hw, hw, generator=gen)}
    steps, gs, isa, ibg = 3, 6.0, 0.5, 0.5
    guidance = {"vae": OracleVAEDecoder(TINY_VAE_CONFIG, vsd), "scaling": TINY_VAE_CONFIG["scaling_factor"]}
    vae = VaeDecoder(TINY_VAE_CONFIG, hw, hw, device=0, state_dict=vsd)
    if xl:
        from rich_text_to_image_amd.region_diffusion_sdxl import RegionDiffusionXL
        sched = OracleEuler(); sched.set_timesteps(steps)
        tid = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]])
        ref = region_loop.rich_loop_xl(OracleUNet(cfg, sd), OracleEuler(), inp["embeds"], inp["pooled"], tid, masks, lat * sched.init_noise_sigma,
                                       steps, gs, tfd, isa, ibg, use_guidance=True, guidance=guidance)
        mdl = RegionDiffusionXL(device=0, unet_state_dict=sd, config=cfg, vae=vae, vae_scaling_factor=TINY_VAE_CONFIG["scaling_factor"])
        mdl.masks = masks
        out = mdl.sample(prompt=None, height=8 * hw, width=8 * hw, num_inference_steps=steps, guidance_scale=gs, latents=lat.clone(),
                         prompt_embeds=inp["embeds"][1:], negative_prompt_embeds=inp["embeds"][:1], pooled_prompt_embeds=inp["pooled"][1:],
                         negative_pooled_prompt_embeds=inp["pooled"][:1], output_type="latent", run_rich_text=True, text_format_dict=tfd,
                         use_guidance=True, inject_selfattn=isa, inject_background=ibg).images
    else:
        from rich_text_to_image_amd.region_diffusion import RegionDiffusion
        ref = region_loop.rich_loop_sd(OracleUNet(cfg, sd), OraclePNDM(), inp["embeds"], masks, lat, steps, gs, tfd, isa, ibg,
                                       use_guidance=True, guidance=guidance)
        mdl = RegionDiffusion(0, unet_state_dict=sd, config=cfg, vae=vae)
        mdl.masks = masks
        out = mdl.produce_latents(inp["embeds"], num_inference_steps=steps, guidance_scale=gs, latents=lat.clone(), text_format_dict=tfd,
                                  use_guidance=True, inject_selfattn=isa, inject_background=ibg)
        # without guidance the result must differ: the guidance step really ran
        plain = mdl.produce_latents(inp["embeds"], num_inference_steps=steps, guidance_scale=gs, latents=lat.clone(), text_format_dict=tfd,
                                    use_guidance=False, inject_selfattn=isa, inject_background=ibg)
        assert rel_l2(plain, out) > 1e-4
    r = rel_l2(out, ref)
    print(f"colour-guided rich loop ({'xl' if xl else 'sd'}) vs oracle: rel-L2 {r:.3e}")
    assert r < 3e-2


def test_many_regions_grow_the_engine():
    """Seven attributed spans + base = 8 region prompts => 11 batched forwards per injected step: the facade rebuilds its engine
    with room for them (the default holds 8 streams / 8 prompts) and still matches the oracle loop."""
    from oracle import region_loop
    from oracle.schedulers import OracleEuler
    from oracle.unet import OracleUNet
    from rich_text_to_image_amd.region_diffusion_sdxl import RegionDiffusionXL
    cfg, hw, R, steps = TINY_XL_CONFIG, 128, 8, 2
    sd = random_state_dict(cfg, seed=21)
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(R + 1, 77, cfg["cross_attention_dim"], generator=g)
    pooled = torch.randn(R + 1, 32, generator=g)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    m = torch.softmax(torch.randn(R, 1, hw, hw, generator=g) * 2, 0).repeat(1, 4, 1, 1)
    masks = [m[r:r + 1] for r in range(R)]
    tfd = {"word_pos": torch.tensor([2, 3]), "font_size": torch.tensor([2.0, -1.5])}
    sched = OracleEuler(); sched.set_timesteps(steps)
    tid = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]])
    ref = region_loop.rich_loop_xl(OracleUNet(cfg, sd), OracleEuler(), emb, pooled, tid, masks, lat * sched.init_noise_sigma, steps, 5.0, tfd, 1.0, 0.0)
    mdl = RegionDiffusionXL(device=0, unet_state_dict=sd, config=cfg)
    assert mdl.unet.max_streams == 8
    mdl.masks = masks
    out = mdl.sample(prompt=None, height=8 * hw, width=8 * hw, num_inference_steps=steps, guidance_scale=5.0, latents=lat.clone(),
                     prompt_embeds=emb[1:], negative_prompt_embeds=emb[:1], pooled_prompt_embeds=pooled[1:], negative_pooled_prompt_embeds=pooled[:1],
                     output_type="latent", run_rich_text=True, text_format_dict=tfd, inject_selfattn=1.0, inject_background=0.0).images
    assert mdl.unet.max_streams >= R + 3 and mdl.unet.max_prompts >= R + 1
    r = rel_l2(out, ref)
    print("8 regions (11 streams) vs oracle rel-L2", r)
    assert r < 3e-2
    # a rank != 0 of a seed-parallel launch holds NO state dict ("empty": its weights arrived as the packed arena in one broadcast,
    # sample.py build_model); growing its engine must carry the arena over device to device (the rebuilt engine used to come
    # up unbound and the next forward threw 'weight not bound')
    from rich_text_to_image_amd.launcher import arena_tensor
    rx = RegionDiffusionXL(device=0, unet_state_dict="empty", config=cfg)
    e8 = rx.unet.engine(hw, hw)
    assert rx.unet.max_streams == 8 and e8.weights_missing()[0] > 0
    arena_tensor(e8).copy_(arena_tensor(mdl.unet.engine(hw, hw)))             # what launcher.broadcast_weights does on a receiving rank
    e8.arena_mark_bound()
    rx.masks = masks
    out2 = rx.sample(prompt=None, height=8 * hw, width=8 * hw, num_inference_steps=steps, guidance_scale=5.0, latents=lat.clone(),
                     prompt_embeds=emb[1:], negative_prompt_embeds=emb[:1], pooled_prompt_embeds=pooled[1:], negative_pooled_prompt_embeds=pooled[:1],
                     output_type="latent", run_rich_text=True, text_format_dict=tfd, inject_selfattn=1.0, inject_background=0.0).images
    assert rx.unet.max_streams >= R + 3 and rx.unet.engine(hw, hw).weights_missing()[0] == 0
    assert torch.equal(out2, out)
