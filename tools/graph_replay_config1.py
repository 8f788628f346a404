#!/usr/bin/env python3
"""Is BASELINE config 1 (SD-v1.5 512^2, R = 2, 3 forwards per PLMS iteration: ~264 launches of 15 us on average) bound by the host's launch
rate?  The 21 iterations of the schedule eager, and the same iterations captured into one HIP graph each and replayed back to back.
    python tools/graph_replay_config1.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import masks_for  # noqa: E402
from rich_text_to_image_amd.engine import SD15_CONFIG  # noqa: E402
from rich_text_to_image_amd.region_diffusion import RegionDiffusion  # noqa: E402


def main():
    g = torch.Generator().manual_seed(0)
    R, hw, steps, gs = 2, 64, 20, 8.5
    m = RegionDiffusion(0, unet_state_dict="random0", config=SD15_CONFIG)
    m.masks = masks_for(R, hw, g)
    emb = torch.randn(R + 1, 77, 768, generator=g)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    ref = m.produce_latents(emb, num_inference_steps=steps, guidance_scale=gs, latents=lat.clone()).clone()       # programs the engine (prompts, masks, schedule)
    eng = m.unet.engine(hw, hw, streams=R + 3, prompts=R + 1)
    n = len(m.scheduler.timesteps)

    def reset():
        eng.set_schedule(1, m.scheduler.timesteps.tolist(), m.scheduler.table(), steps)
        eng.set_latents(lat.to("cuda:0"))

    def eager():
        for i in range(n):
            eng.region_step(i, gs, 0.0, 0.0, xl=False, elide=False)
    for _ in range(2):
        reset(); eager()
    eng.synchronize(); torch.cuda.synchronize()
    best_e = 1e9
    for _ in range(5):
        reset(); eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter(); eager(); eng.synchronize(); torch.cuda.synchronize()
        best_e = min(best_e, time.perf_counter() - t0)
    same_e = bool(torch.equal(eng.read_latents(hw, hw).cpu(), ref.cpu()))
    side = torch.cuda.Stream()
    eng.synchronize(); eng.set_stream(side.cuda_stream)
    try:
        reset(); torch.cuda.synchronize()
        graphs = []
        for i in range(n):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=side):
                eng.region_step(i, gs, 0.0, 0.0, xl=False, elide=False)
            graphs.append(gph)
        best_g = 1e9
        for _ in range(5):
            reset(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for gph in graphs:
                gph.replay()
            torch.cuda.synchronize()
            best_g = min(best_g, time.perf_counter() - t0)
        same_g = bool(torch.equal(eng.read_latents(hw, hw).cpu(), ref.cpu()))
    finally:
        eng.synchronize(); eng.set_stream(None)
    print(f"config 1, {n} PLMS iterations: eager {best_e * 1e3:.1f} ms = {n / best_e:.1f} steps/s (latents identical with the facade run: {same_e}) | "
          f"one HIP graph per iteration, replayed {best_g * 1e3:.1f} ms = {n / best_g:.1f} steps/s (identical: {same_g})")


if __name__ == "__main__":
    main()
