"""Kernel profile target: ten colour-guidance calls of the PRECISE SDXL VAE at a 128^2 latent (what config 5 / the end-to-end rich pass
spend 92 ms per step in).   rocprofv3 --kernel-trace --stats -- python tools/vae_precise_profile.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import guidance_dict, random_vae  # noqa: E402
from rich_text_to_image_amd.engine import SDXL_VAE_CONFIG  # noqa: E402

hw = 128
vae = random_vae(SDXL_VAE_CONFIG, hw, hw, precise=True)
g = torch.Generator().manual_seed(3)
tfd = guidance_dict(hw, g, 1, 0.5)
lat = torch.randn(1, 4, hw, hw, generator=g).cuda()
eps = torch.randn(1, 4, hw, hw, generator=g).cuda()
run = lambda: vae.color_guidance(lat, eps, 0.37, hw, hw, tfd["color_obj_atten"], tfd["target_RGB"], 0.5, tfd["color_obj_atten_all"])
for _ in range(2):
    run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
print(f"precise SDXL VAE guidance call: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
