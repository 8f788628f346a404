"""TEST INFRASTRUCTURE: generate tests/golden/*.pt by running the UNMODIFIED reference loops.

Runs only in the build container (needs /root/reference). It drives
  * RegionDiffusion.produce_latents        (/root/reference/models/region_diffusion.py:86-174)
  * RegionDiffusionXL.sample(run_rich_text) (/root/reference/models/region_diffusion_sdxl.py:555-953)
on tiny random-weight UNets (oracle.unet.TINY_*_CONFIG) with the restated schedulers
(oracle/schedulers.py; diffusers itself is not available), and stores inputs + outputs so that the
GPU-box tests can check (a) the oracle restatement and (b) the HIP engine against outputs of the
reference code itself.  Usage:  python -m oracle.make_golden
"""
import os
import sys
import types

import torch

from .refload import load_reference
from .schedulers import OracleEuler, OraclePNDM
from .unet import (TINY_SD_CONFIG, TINY_XL_CONFIG, OracleUNet, random_state_dict, reference_kwargs)
from . import region_loop

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_masks(R, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.softmax(torch.randn(R, 1, h // 4, w // 4, generator=g) * 4, dim=0)
    m = torch.nn.functional.interpolate(m, size=(h, w), mode="bilinear", align_corners=False)
    m = m / (m.sum(0, keepdim=True) + 1e-8)
    return [m[r:r + 1].repeat(1, 4, 1, 1).contiguous() for r in range(R)]


def case_inputs(cfg, xl, R, hw, seed):
    g = torch.Generator().manual_seed(seed)
    D = cfg["cross_attention_dim"]
    inp = {
        "latents": torch.randn(1, 4, hw, hw, generator=g),
        "embeds": torch.randn(R + 1, 77, D, generator=g),
        "masks": make_masks(R, hw, hw, seed + 1),
        "word_pos": torch.tensor([3, 5], dtype=torch.long),
        "font_size": torch.tensor([4.0, -2.0]),
    }
    if xl:
        inp["pooled"] = torch.randn(R + 1, 32, generator=g)
        inp["time_ids"] = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]])
    return inp


def run_reference_sd(mods, unet, inp, steps, gs, inject_selfattn, inject_background):
    RD = mods["region_diffusion"].RegionDiffusion
    m = RD.__new__(RD)
    torch.nn.Module.__init__(m)
    m.device = torch.device("cpu")
    m.unet = unet
    m.scheduler = OraclePNDM()
    m.masks = inp["masks"]
    unet.in_channels = 4
    tfd = {"word_pos": inp["word_pos"], "font_size": inp["font_size"]}
    with torch.no_grad():
        return m.produce_latents(inp["embeds"], height=inp["latents"].shape[2] * 8, width=inp["latents"].shape[3] * 8,
                                 num_inference_steps=steps, guidance_scale=gs, latents=inp["latents"].clone(),
                                 text_format_dict=tfd, inject_selfattn=inject_selfattn,
                                 inject_background=inject_background)


def run_reference_xl(mods, unet, inp, steps, gs, inject_selfattn, inject_background):
    XL = mods["region_diffusion_sdxl"].RegionDiffusionXL
    m = XL.__new__(XL)
    m.unet = unet
    m.scheduler = OracleEuler()
    m.device_type = "cpu"
    m.vae_scale_factor = 8
    m.default_sample_size = inp["latents"].shape[2]
    m.register_to_config(force_zeros_for_empty_prompt=True)
    m.tokenizer = m.tokenizer_2 = m.text_encoder = None
    m.text_encoder_2 = types.SimpleNamespace(config=types.SimpleNamespace(projection_dim=32))
    dummy_vae = types.SimpleNamespace(
        to=lambda **k: None,
        decoder=types.SimpleNamespace(mid_block=types.SimpleNamespace(attentions=[types.SimpleNamespace(processor=None)])))
    m.vae = dummy_vae
    m.masks = inp["masks"]
    m.check_inputs = lambda *a, **k: None
    sched = m.scheduler
    sched.set_timesteps(steps)
    lat = inp["latents"].clone()          # prepare_latents multiplies by init_noise_sigma (xl.py:536)
    tfd = {"word_pos": inp["word_pos"], "font_size": inp["font_size"]}
    hw = inp["latents"].shape[2] * 8
    out = m.sample(prompt=None, height=hw, width=hw, num_inference_steps=steps, guidance_scale=gs,
                   latents=lat, prompt_embeds=inp["embeds"][1:], negative_prompt_embeds=inp["embeds"][:1],
                   pooled_prompt_embeds=inp["pooled"][1:], negative_pooled_prompt_embeds=inp["pooled"][:1],
                   output_type="latent", run_rich_text=True, text_format_dict=tfd,
                   inject_selfattn=inject_selfattn, inject_background=inject_background,
                   original_size=(hw, hw), target_size=(hw, hw))
    return out.images


def main():
    os.makedirs(OUT, exist_ok=True)
    mods = load_reference()
    U = mods["unet_2d_condition"].UNet2DConditionModel
    AP = mods["attention_processor"]
    cases = [
        # name, cfg, xl, R, latent hw, steps, cfg scale, inject_selfattn, inject_background
        ("tiny_sd_plms", TINY_SD_CONFIG, False, 3, 64, 6, 7.5, 0.5, 0.5),
        ("tiny_sd_noinject", TINY_SD_CONFIG, False, 2, 64, 4, 8.5, 0.0, 0.0),
        ("tiny_xl_euler", TINY_XL_CONFIG, True, 3, 128, 6, 5.0, 0.5, 0.5),
        ("tiny_xl_bgonly", TINY_XL_CONFIG, True, 2, 128, 4, 7.5, 0.0, 0.5),
    ]
    for name, cfg, xl, R, hw, steps, gs, isa, ibg in cases:
        wseed = 11
        sd = random_state_dict(cfg, seed=wseed)
        ref = U(**reference_kwargs(cfg))
        ref.load_state_dict(sd)
        ref.eval()
        inp = case_inputs(cfg, xl, R, hw, seed=5)
        if xl:
            ref_out = run_reference_xl(mods, ref, inp, steps, gs, isa, ibg)
            sched = OracleEuler()
            sched.set_timesteps(steps)
            lat0 = inp["latents"] * sched.init_noise_sigma
            trace = []
            orc = region_loop.rich_loop_xl(OracleUNet(cfg, sd), OracleEuler(), inp["embeds"], inp["pooled"],
                                           inp["time_ids"], inp["masks"], lat0, steps, gs,
                                           {"word_pos": inp["word_pos"], "font_size": inp["font_size"]},
                                           isa, ibg, trace=trace)
        else:
            ref_out = run_reference_sd(mods, ref, inp, steps, gs, isa, ibg)
            trace = []
            orc = region_loop.rich_loop_sd(OracleUNet(cfg, sd), OraclePNDM(), inp["embeds"], inp["masks"],
                                           inp["latents"], steps, gs,
                                           {"word_pos": inp["word_pos"], "font_size": inp["font_size"]},
                                           isa, ibg, trace=trace)
        err = (ref_out - orc).abs().max().item()
        print(f"{name}: reference-vs-oracle max|diff| = {err:.3e}  (out std {ref_out.std().item():.3f})")
        assert err < 5e-4, name
        # single UNet forward golden (reference module output)
        with torch.no_grad():
            added = {"text_embeds": inp["pooled"][1:2], "time_ids": inp["time_ids"]} if xl else None
            unet_out = ref(inp["latents"], torch.tensor(481), encoder_hidden_states=inp["embeds"][1:2],
                           added_cond_kwargs=added)["sample"]
        torch.save({
            "name": name, "xl": xl, "R": R, "steps": steps, "guidance_scale": gs,
            "inject_selfattn": isa, "inject_background": ibg, "weight_seed": wseed,
            "weight_abs_sum": float(sum(v.abs().sum() for v in sd.values())),
            "inputs": {k: (v if not isinstance(v, list) else torch.cat(v)[:, :1].clone()) for k, v in inp.items()},
            "reference_final_latents": ref_out.clone(),
            "reference_unet_t481": unet_out.clone(),
            "oracle_trace_last": trace[-1].clone(),
        }, os.path.join(OUT, name + ".pt"))

    # operator-level golden: the reference Attention module (font-size + injection identities)
    torch.manual_seed(3)
    attn = AP.Attention(query_dim=64, cross_attention_dim=48, heads=2, dim_head=32)
    sattn = AP.Attention(query_dim=64, heads=2, dim_head=32)
    x = torch.randn(2, 256, 64)
    ctx = torch.randn(2, 77, 48)
    fs = {"word_pos": torch.tensor([2, 9, 30]), "font_size": torch.tensor([3.0, -1.5, 0.25])}
    with torch.no_grad():
        y_plain, _ = attn(x, encoder_hidden_states=ctx)
        y_fs, (_, p_fs) = attn(x, None, fs, encoder_hidden_states=ctx)
        y_self, (pavg, p_self) = sattn(x)
        x2 = torch.randn(2, 256, 64)
        y_inj, _ = sattn(x2, p_self)
    torch.save({
        "cross_sd": {k: v.clone() for k, v in attn.state_dict().items()},
        "self_sd": {k: v.clone() for k, v in sattn.state_dict().items()},
        "x": x, "ctx": ctx, "x2": x2, "word_pos": fs["word_pos"], "font_size": fs["font_size"],
        "y_plain": y_plain, "y_fs": y_fs, "p_fs_rowsum": p_fs.sum(-1), "y_self": y_self,
        "p_self_avg": pavg, "y_inj": y_inj,
    }, os.path.join(OUT, "attention_ops.pt"))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
