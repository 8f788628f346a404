"""Colour-guidance call (VAE decoder forward + input gradient) timing: SD VAE at 64^2, SDXL VAE at 128^2 single-pass and precise; with the
stride-1 convolutions on the gemm16 main loop (default) and on the patch kernel (rt_op_gemm_debug bit 3).  python tools/vae_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import guidance_dict, random_vae  # noqa: E402
from rich_text_to_image_amd.engine import SD_VAE_CONFIG, SDXL_VAE_CONFIG, load_library  # noqa: E402

lib = load_library()
for name, cfg, hw, precise in (("SD 64^2", SD_VAE_CONFIG, 64, False), ("SDXL 128^2", SDXL_VAE_CONFIG, 128, False), ("SDXL 128^2 precise", SDXL_VAE_CONFIG, 128, True)):
    vae = random_vae(cfg, hw, hw, precise=precise)
    g = torch.Generator().manual_seed(3)
    tfd = guidance_dict(hw, g, 1, 0.5)
    lat = torch.randn(1, 4, hw, hw, generator=g).cuda()
    eps = torch.randn(1, 4, hw, hw, generator=g).cuda()
    for flags in (0, 8):
        lib.rt_op_gemm_debug(flags)

        def run():
            vae.color_guidance(lat, eps, 0.37, hw, hw, tfd["color_obj_atten"], tfd["target_RGB"], 0.5, tfd["color_obj_atten_all"])
        for _ in range(2):
            run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        print(f"{name:20s} debug flags {flags}: {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms per guidance call")
    lib.rt_op_gemm_debug(0)
    vae.close()
