"""Secondary measurements through the drop-in facade classes, random-init weights of the true
architectures, synthetic inputs.  One JSON line per configuration; `bench.py` stays the headline (config 3) line.
    python tools/bench_configs.py [--configs 1,2,5] > profiles/rN_configs.jsonl"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rich_text_to_image_amd.engine import SD15_CONFIG, SD_VAE_CONFIG, SDXL_CONFIG, SDXL_VAE_CONFIG, VaeDecoder  # noqa: E402


def random_vae(cfg, h, w, seed=1, precise=False):
    vae = VaeDecoder(cfg, h, w, device=0, precise=precise)
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    sd = {}
    for name, shape in vae.weight_table():
        if name.endswith(".weight") and len(shape) >= 2:
            sd[name] = (torch.rand(shape, generator=g, device="cuda:0") * 2 - 1) / math.sqrt(math.prod(shape[1:]))
        elif name.endswith(".weight"):
            sd[name] = 1.0 + 0.1 * (torch.rand(shape, generator=g, device="cuda:0") * 2 - 1)
        else:
            sd[name] = 0.05 * (torch.rand(shape, generator=g, device="cuda:0") * 2 - 1)
    vae.load_state_dict(sd)
    return vae


def masks_for(R, hw, g):
    m = torch.softmax(torch.randn(R, 1, hw // 4, hw // 4, generator=g) * 4, dim=0)
    m = torch.nn.functional.interpolate(m, size=(hw, hw), mode="bilinear", align_corners=False)
    m = (m / (m.sum(0, keepdim=True) + 1e-8)).repeat(1, 4, 1, 1)
    return [m[r:r + 1] for r in range(R)]


def guidance_dict(hw, g, n_colors=2, weight=1.0):
    cm = [torch.nn.functional.interpolate(torch.rand(1, 1, hw // 4, hw // 4, generator=g), size=(8 * hw, 8 * hw), mode="bicubic").clamp(0, 1)
          .repeat(1, 4, 1, 1) for _ in range(n_colors)]
    return {"word_pos": None, "font_size": None, "target_RGB": [torch.rand(1, 3, 1, 1, generator=g) for _ in range(n_colors)],
            "guidance_start_step": 999, "color_guidance_weight": weight, "color_obj_atten": cm,
            "color_obj_atten_all": torch.rand(1, 4, hw, hw, generator=g).clamp(0, 1)}


STEPS_OVERRIDE = 0     # bench.py --config N --steps K
WARM_STEPS = 3


def timed(fn, warm):
    warm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def config1():
    from rich_text_to_image_amd.region_diffusion import RegionDiffusion
    g = torch.Generator().manual_seed(0)
    R, hw, steps = 2, 64, STEPS_OVERRIDE or 20
    m = RegionDiffusion(0, unet_state_dict="random0", config=SD15_CONFIG)
    m.masks = masks_for(R, hw, g)
    emb = torch.randn(R + 1, 77, 768, generator=g)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    run = lambda n: m.produce_latents(emb, num_inference_steps=n, guidance_scale=8.5, latents=lat.clone())
    out, dt = timed(lambda: run(steps), lambda: run(WARM_STEPS))
    it = steps + 1
    return dict(config=1, workload=f"SD-v1.5 RegionDiffusion 512^2, R=2, {steps}-step PLMS ({steps + 1} iterations), CFG 8.5, 3 forwards/iteration",
                iterations=it, seconds=dt, value=it / dt, unit="steps/s", tflop_per_iteration=3 * 0.8033, finite=bool(torch.isfinite(out).all()))


def config2():
    from rich_text_to_image_amd.region_diffusion import RegionDiffusion
    g = torch.Generator().manual_seed(1)
    R, hw, steps = 4, 64, STEPS_OVERRIDE or 50
    vae = random_vae(SD_VAE_CONFIG, hw, hw)
    m = RegionDiffusion(0, unet_state_dict="random0", config=SD15_CONFIG, vae=vae)
    m.masks = masks_for(R, hw, g)
    emb = torch.randn(R + 1, 77, 768, generator=g)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    tfd = guidance_dict(hw, g, 2, 1.0)
    run = lambda n: m.produce_latents(emb, num_inference_steps=n, guidance_scale=7.5, latents=lat.clone(), text_format_dict=tfd, use_guidance=True)
    out, dt = timed(lambda: run(steps), lambda: run(WARM_STEPS))
    it = steps + 1
    return dict(config=2, workload=f"SD-v1.5 512^2, R=4, {steps}-step PLMS ({steps + 1} iterations), colour guidance weight 1 on 2 regions (VAE decoder forward + input gradient every iteration)",
                iterations=it, seconds=dt, value=it / dt, unit="steps/s", tflop_per_iteration=5 * 0.8033 + 5.0, finite=bool(torch.isfinite(out).all()))


def config5():
    from rich_text_to_image_amd.region_diffusion_sdxl import RegionDiffusionXL
    g = torch.Generator().manual_seed(2)
    R, hw, steps = 4, 128, STEPS_OVERRIDE or 50
    # the SDXL pipeline of the reference runs the guidance VAE in fp32 (xl.py:856): the precise engine (3 bf16 MFMA passes over hi/lo pairs)
    vae = random_vae(SDXL_VAE_CONFIG, hw, hw, precise=True)
    m = RegionDiffusionXL(device=0, unet_state_dict="random0", config=SDXL_CONFIG, vae=vae, vae_scaling_factor=SDXL_VAE_CONFIG["scaling_factor"])
    m.masks = masks_for(R, hw, g)
    emb, pooled = torch.randn(R + 1, 77, 2048, generator=g), torch.randn(R + 1, 1280, generator=g)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    tfd = guidance_dict(hw, g, 1, 0.5)
    run = lambda n: m.sample(prompt=None, height=8 * hw, width=8 * hw, num_inference_steps=n, guidance_scale=7.5, latents=lat.clone(),
                             prompt_embeds=emb[1:], negative_prompt_embeds=emb[:1], pooled_prompt_embeds=pooled[1:],
                             negative_pooled_prompt_embeds=pooled[:1], output_type="latent", run_rich_text=True, text_format_dict=tfd,
                             use_guidance=True, inject_selfattn=0.0, inject_background=0.5).images
    out, dt = timed(lambda: run(steps), lambda: run(WARM_STEPS))
    # informational (round 6): the same loop with the guidance pass alone on a one-pass bf16 VAE engine (RegionDiffusionXL.guidance_vae);
    # NOT the line's value - the reference guides through an fp32 VAE (xl.py:856)
    m.guidance_vae = random_vae(SDXL_VAE_CONFIG, hw, hw, precise=False)
    out2, dt2 = timed(lambda: run(steps), lambda: run(WARM_STEPS))
    m.guidance_vae.close(); m.guidance_vae = None
    fastg = dict(value=steps / dt2, unit="steps/s", finite=bool(torch.isfinite(out2).all()),
                 latent_rel_l2_vs_precise_guidance=float(((out2 - out).float().norm() / out.float().norm()).item()),
                 note="colour guidance on a one-pass bf16 VaeDecoder (sample.py --guidance_precision bf16), final decode unchanged; informational")
    return dict(config=5, one_pass_guidance=fastg, workload=f"SDXL 1024^2, R=4 (footnote+style+colour+base), {steps}-step Euler, CFG 7.5, colour guidance on 1 region through the fp32-class (precise) VAE, inject_background=0.5, 7 forwards/step",
                iterations=steps, seconds=dt, value=steps / dt, unit="steps/s", tflop_per_iteration=7 * 6.7612 + 21.2, finite=bool(torch.isfinite(out).all()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,5")
    ap.add_argument("--steps", type=int, default=0, help="override the step count (profiling runs)")
    a = ap.parse_args()
    if a.steps:
        STEPS_OVERRIDE = a.steps
    for c in a.configs.split(","):
        r = {"1": config1, "2": config2, "5": config5}[c]()
        r["tflops"] = r["tflop_per_iteration"] * r["value"]
        r["data"] = "synthetic, random-init weights"
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()
