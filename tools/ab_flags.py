#!/usr/bin/env python3
"""Same-box, same-process A/B of rt_op_gemm_debug switches on the headline step (config 3: SDXL 1024^2, R = 4, 7 streams).

  python tools/ab_flags.py --flags 0 1024 2048 --rounds 3 --steps 20

ONE engine (random-init SDXL-base weights, bench.py's synthetic request), the timed loop of bench.py (`--steps` steps strided over the
50-step schedule so injected and non-injected steps are mixed as in the schedule), repeated `--rounds` times for every flag value in
interleaved order (A B C A B C ...), so clock / thermal drift of the box hits every variant alike.  Prints one JSON line per variant:
mean / min ms per step and, with --profile, the per-class event totals of one injected + one non-injected step.  The final latents of
every variant are compared with those of the first one (`max_abs_diff_vs_first`): a switch that only changes the store instruction must
read 0.0."""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (synth_inputs, euler_tables)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, nargs="+", default=[0])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG, load_library
    lib = load_library()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    R, hw, nsched, gs, isa, ibg = 4, 128, 50, 5.0, 0.5, 0.0
    eng = Engine(SDXL_CONFIG, hw, hw, device=0, max_streams=8, max_prompts=8)
    g = torch.Generator(device=dev).manual_seed(0)
    for name, shape in eng.weight_table():
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = (torch.rand(shape, generator=g, device=dev) * 2 - 1) / math.sqrt(fan_in)
        elif name.endswith(".weight"):
            t = 1.0 + 0.1 * (torch.rand(shape, generator=g, device=dev) * 2 - 1)
        else:
            t = 0.05 * (torch.rand(shape, generator=g, device=dev) * 2 - 1)
        eng.bind_weight(name, t)
        eng.synchronize()
        del t
    inp = bench.synth_inputs(1000, R, hw, dev)
    ts, sig, init_sigma = bench.euler_tables(nsched)
    eng.set_prompts(inp["emb"], inp["pooled"], inp["tid"])
    eng.set_masks(inp["masks"])
    eng.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
    lat0 = (inp["lat"] * init_sigma).to(dev)

    def reset():
        eng.set_schedule(0, ts, sig, nsched)
        eng.set_latents(lat0)

    def sched_index(i, k):
        return (i * nsched) // k if k < nsched else i % nsched

    def run(k):
        for i in range(k):
            eng.region_step(sched_index(i, k), gs, isa, ibg, xl=True, elide=False)

    times = {f: [] for f in args.flags}
    finals = {}
    for f in args.flags:                                     # warm every variant's kernels once
        lib.rt_op_gemm_debug(f)
        reset(); run(args.warmup); eng.region_step(nsched - 1, gs, isa, ibg, xl=True, elide=False); eng.synchronize()
    for r in range(args.rounds):
        for f in args.flags:
            lib.rt_op_gemm_debug(f)
            reset(); eng.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            eng.synchronize(); torch.cuda.synchronize()
            times[f].append((time.perf_counter() - t0) / args.steps * 1e3)
            if r == 0:
                finals[f] = eng.read_latents(hw, hw).clone()
    first = finals[args.flags[0]]
    for f in args.flags:
        out = dict(flags=f, ms_per_step_mean=sum(times[f]) / len(times[f]), ms_per_step_min=min(times[f]), rounds=times[f],
                   finite=bool(torch.isfinite(finals[f]).all()), max_abs_diff_vs_first=float((finals[f] - first).abs().max()))
        if args.profile:
            lib.rt_op_gemm_debug(f)
            reset()
            per = []
            for i_prof in (0, nsched - 1):
                eng.profile_enable(True)
                eng.region_step(i_prof, gs, isa, ibg, xl=True, elide=False)
                per.append(eng.profile_read())
                eng.profile_enable(False)
            out["classes_ms_per_2_steps"] = {k: round(per[0][k]["total_ms"] + per[1][k]["total_ms"], 3) for k in per[0]}
        print(json.dumps(out), flush=True)
    lib.rt_op_gemm_debug(0)


if __name__ == "__main__":
    main()
