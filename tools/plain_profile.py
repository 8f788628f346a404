#!/usr/bin/env python3
"""The plain-text pass alone (SDXL 1024^2, batch-2 CFG forward per step, sample.py:59-75) for a rocprofv3 kernel trace:
  rocprofv3 --kernel-trace --stats -d <dir> -- python tools/plain_profile.py [--steps 12] [--capture]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import end_to_end as e2e  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--capture", action="store_true")
    args = ap.parse_args()
    from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
    torch.cuda.set_device(0)
    hw = 128
    eng = Engine(SDXL_CONFIG, hw, hw, device=0, max_streams=8, max_prompts=8)
    eng.init_random_weights(0)
    inp = bench.synth_inputs(1000, 4, hw, "cuda:0")
    ts, sig, init_sigma = e2e.euler_tables(41)
    eng.set_prompts(inp["emb"][[0, -1]], inp["pooled"][[0, -1]], inp["tid"])
    rec = e2e.recorded_modules(eng)
    for n, _, _ in eng.attn_modules():
        eng.attn_store_enable(n, 1 if (args.capture and n in rec) else 0)
    lat0 = (inp["lat"] * init_sigma).to("cuda:0")
    eng.set_schedule(0, ts, sig, 41); eng.set_latents(lat0); eng.attn_store_reset()
    for i in range(3):
        eng.plain_step(i, 5.0)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.plain_step(3 + i, 5.0)
    eng.synchronize(); torch.cuda.synchronize()
    print(f"plain pass: {(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms per step (capture {'on' if args.capture else 'off'})")


if __name__ == "__main__":
    main()
