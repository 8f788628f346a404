#!/usr/bin/env python3
"""Headline benchmark: denoising steps/sec of the rich-text (region-diffusion) loop.

Workload = BASELINE.json configs[2] ("SDXL RegionDiffusionXL 1024x1024, 4 regions, inject_selfattn=0.5,
50 steps, 1xMI355X"), the configuration the metric is quoted on:
  R = 4 masks (3 region prompts + base), F = R+3 = 7 UNet forwards per step (uncond, base with font-size
  softmax, uncond_ref, text_ref, 3 region forwards with self-attention / ResNet-feature injection while
  t > 500), CFG 5.0, 50-step Euler schedule, latents 128x128, SDXL-base architecture with random-init
  weights and synthetic conditioning (no checkpoints offline).
A "step" is one iteration of the loop at models/region_diffusion_sdxl.py:779 (all F forwards, mask combine,
CFG, scheduler step).  Every forward the reference executes is executed (no dead-forward elision).

  python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU under torch.distributed.run - either launched that way by the caller (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when `python bench.py --gpus N` is run bare, by re-executing itself under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (launcher.self_launch).  Each rank runs an independent image (own seed,
latents, masks, prompts) after ONE broadcast of the packed weight arena (RCCL); no per-step collectives;
value = N*K / max-over-ranks time ("weak" scaling).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SDXL_FWD_GFLOP = 6761.2          # SURVEY.md section 8 [probe]: FLOPs of one batch-1 SDXL UNet forward at 128x128
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PARITY_TOL = 1.5e-2              # rel-L2 of one full-architecture UNet forward vs the fp32 oracle (tests/test_fullsize_gpu.py)


def synth_inputs(seed, R, hw, device):
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(R + 1, 77, 2048, generator=g)
    pooled = torch.randn(R + 1, 1280, generator=g)
    tid = torch.tensor([[hw * 8.0, hw * 8.0, 0.0, 0.0, hw * 8.0, hw * 8.0]])
    m = torch.softmax(torch.randn(R, 1, hw // 4, hw // 4, generator=g) * 4, dim=0)
    m = torch.nn.functional.interpolate(m, size=(hw, hw), mode="bilinear", align_corners=False)
    m = (m / (m.sum(0, keepdim=True) + 1e-8)).repeat(1, 4, 1, 1)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    return dict(emb=emb.to(device), pooled=pooled.to(device), tid=tid, masks=m.to(device), lat=lat)


def euler_tables(n):
    """EulerDiscreteScheduler tables (restated diffusers 0.18.2, see oracle/schedulers.py) computed inline so the
    product path does not import the oracle."""
    import numpy as np
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, 0)
    train_sig = (((1 - ac) / ac) ** 0.5).numpy().astype("float64")
    ts = (np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype("float32") + 1
    sig = np.concatenate([np.interp(ts, np.arange(0, 1000), train_sig), [0.0]]).astype("float32")
    return ts.tolist(), sig.tolist(), float((sig.max() ** 2 + 1) ** 0.5)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def pick_cpu_threads(threads_max):
    """Thread count for the fp32 oracle on this host: the fastest of {16, 32, 64, 128, all physical cores} on the operators one ResNet
    block + one transformer block of the SDXL UNet are made of, at their 1280-channel / 32x32 shapes (3x3 convolution 1280 -> 1280,
    the GEGLU and projection matmuls, 20-head self-attention over 1024 tokens, GroupNorm / LayerNorm): oversubscribed NUMA hosts are
    slower with every hardware thread than with a subset, and a conv-heavy forward need not prefer what one big matmul prefers."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    xc, wc = torch.randn(1, 1280, 32, 32, generator=g), torch.randn(1280, 1280, 3, 3, generator=g) * 0.01
    xt, w1, w2, wq = torch.randn(1024, 1280, generator=g), torch.randn(10240, 1280, generator=g) * 0.02, torch.randn(1280, 5120, generator=g) * 0.02, torch.randn(1280, 1280, generator=g) * 0.02
    def block():
        h = F.conv2d(F.silu(F.group_norm(xc, 32)), wc, padding=1)
        h = F.conv2d(F.silu(F.group_norm(h, 32)), wc, padding=1)
        y = F.layer_norm(xt, (1280,))
        q = (y @ wq.t()).reshape(1024, 20, 64).transpose(0, 1)
        a = torch.softmax(q @ q.transpose(1, 2) * 0.125, -1) @ q
        y = y + a.transpose(0, 1).reshape(1024, 1280) @ wq.t()
        u = F.layer_norm(y, (1280,)) @ w1.t()
        y = y + (u[:, :5120] * F.gelu(u[:, 5120:])) @ w2.t()
        return h.sum() + y.sum()
    cand = sorted({t for t in (16, 32, 64, 128, threads_max) if t <= threads_max})
    probe = {}
    with torch.no_grad():
        for nt in cand:
            torch.set_num_threads(nt)
            block()
            t0 = time.perf_counter(); block(); block()
            probe[nt] = (time.perf_counter() - t0) / 2
    best = min(probe, key=probe.get)
    return best, cand, {str(k): round(v, 4) for k, v in probe.items()}


def cpu_baseline(sd_cpu, threads, x, ctx, pooled, tid, t, full_step=True, all_ctx=None, all_pooled=None):
    """Reference-equivalent CPU path: the fp32 oracle restatement of the reference UNet (oracle/unet.py, pinned
    against the unmodified reference) timed on the host cores of this box.  Bounded sample: ONE batch-1 SDXL
    UNet forward at 128x128 (the step is 7 such forwards + negligible elementwise work), on the SAME latents /
    prompt / timestep as one stream of the engine, so its output doubles as the full-architecture parity check."""
    from oracle.unet import SDXL_CONFIG, OracleUNet
    threads_max = threads
    threads, cand, probe = pick_cpu_threads(threads)
    torch.set_num_threads(threads)
    o = OracleUNet(SDXL_CONFIG, sd_cpu)
    added = {"text_embeds": pooled, "time_ids": tid}
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = o.forward(x, t, ctx, added)
        dt = time.perf_counter() - t0
        if full_step:
            # ONE FULL STEP (whatever --steps says) = the 7 batch-1 forwards of a config-3 iteration (uncond, base, uncond_ref,
            # text_ref, 3 regions: prompts 0, R, 0, R, 1, 2, 3), timed as a whole instead of one forward x 7
            prompts = [0, all_ctx.shape[0] - 1, 0, all_ctx.shape[0] - 1, 1, 2, 3]
            t1 = time.perf_counter()
            for pi in prompts:
                o.forward(x, t, all_ctx[pi:pi + 1], {"text_embeds": all_pooled[pi:pi + 1], "time_ids": tid})
            step_s = time.perf_counter() - t1
            return dict(value=1.0 / step_s, unit="steps/s", cores=threads, cores_physical=threads_max,
                        cores_note=f"{threads} of {threads_max} physical cores: the fastest of {cand} on one ResNet block + one transformer block of the oracle's operators (seconds per probe: {probe})",
                        kind="port", cpu=cpu_model_name(),
                        sample=f"1 full config-3 step = 7 batch-1 SDXL UNet forwards (fp32 oracle, {step_s:.1f} s; hooks / mask combine / Euler update are negligible beside them)",
                        forward_seconds=dt, step_seconds=step_s), ref
    return dict(value=1.0 / (7 * dt), unit="steps/s", cores=threads, cores_physical=threads_max,
                cores_note=f"{threads} of {threads_max} physical cores: the fastest of {cand} on one ResNet block + one transformer block of the oracle's operators (seconds per probe: {probe})",
                kind="port", cpu=cpu_model_name(),
                sample=f"1 batch-1 SDXL UNet forward (fp32 oracle, {dt:.2f} s) x 7 forwards/step extrapolated",
                forward_seconds=dt), ref


def physical_cores():
    threads = max(1, min(os.cpu_count() or 1, 256))
    try:
        phys = len({l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("core id")}) * \
            len({l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("physical id")})
        if phys > 0:
            threads = min(threads, phys)
    except Exception:
        pass
    return threads


# (latents rel-L2 at any recorded iteration, min PSNR [dB], max |d| in uint8 levels, error RELATIVE TO THE ACCUMULATED UPDATE ||lat_k - lat_0|| at any
#  recorded iteration) - the tolerances of tests/test_fullschedule_gpu.py (TOL / UPDATE_TOL).  The update-relative bound is the discriminating
#  one: with seeded random weights the sigma-scaled SDXL latents stay ~95 % start noise, so the latent-relative figure flatters by 2.5 - 3x.
PIXEL_TOL = {"config1": (1.5e-2, 46.0, 8, 2.0e-2), "config2": (1.5e-2, 46.0, 8, 2.0e-2), "config2_50": (1.5e-2, 46.0, 8, 2.0e-2),
             "config3": (1.5e-2, 46.0, 8, 3.0e-2), "config3_50": (1.5e-2, 46.0, 8, 3.0e-2), "config3_unit": (8.0e-2, 34.0, 96, 8.0e-2),
             "config5": (2.5e-2, 46.0, 8, 4.5e-2), "config5_50": (2.5e-2, 46.0, 8, 4.5e-2)}


def pixel_parity(case, also=()):
    """The engine over a FULL schedule against the committed trajectory of the fp32 CPU oracle (tests/golden/fullschedule/<case>.pt,
    oracle/make_fullsize_golden.py): seeded weights regenerated here, the facade classes run the schedule, latents per recorded loop
    iteration and the final uint8 image are compared (tools/fullschedule_check.py).  A checker leg, outside every timed region.
    `also`: further cases on the same model (same weight seeds), reported under "also"."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fullschedule_check as fc
    if fc.load_golden(case) is None:
        return {"error": f"tests/golden/fullschedule/{case}.pt missing"}
    mdl, fp = fc.build_model(case)

    def one(name):
        r = fc.compare(name, mdl, fp)
        t_lat, t_psnr, t_max, t_upd = PIXEL_TOL[name]
        curve, pix = r["latent_rel_l2_by_iteration"], r["pixels_vs_oracle_image"]
        ucurve = r["latent_update_rel_l2_by_iteration"]
        c = r["case"]
        return dict(case=name, schedule=f"{c['model']} latent {c['hw']}x{c['hw']}, R={c['R']}, {c['steps']} scheduler steps = {max(curve)} loop iterations, CFG {c['gs']}, "
                                         f"inject_selfattn {c['isa']}, inject_background {c['ibg']}, colour guidance {c['guided']}",
                    latent_rel_l2_by_iteration={str(k): v for k, v in curve.items()}, latent_rel_l2_final=curve[max(curve)],
                    update_rel_l2_by_iteration={str(k): v for k, v in ucurve.items()}, update_rel_l2_final=ucurve[max(ucurve)], update_rel_l2_max=max(ucurve.values()),
                    update_over_latents_final=r["update_over_latents_final"], latent_abs_rms_final=list(r["latent_abs_rms_by_iteration"].values())[-1],
                    psnr_db=pix["psnr_db"], mean_abs_u8=pix["mean_abs"], max_abs_u8=pix["max_abs"], within_1_level=pix["within_1"],
                    decoder_only_psnr_db=r["decoder_only"]["psnr_db"],
                    tol=dict(latent_rel_l2=t_lat, psnr_db=t_psnr, max_abs_u8=t_max, update_rel_l2=t_upd),
                    ok=bool(max(curve.values()) < t_lat and max(ucurve.values()) < t_upd and pix["psnr_db"] > t_psnr and pix["max_abs"] <= t_max),
                    reference="fp32 CPU oracle trajectory + oracle VAE decode (oracle/region_loop.py, pinned to the unmodified reference loops)")
    try:
        out = one(case)
        extra = {n: one(n) for n in also if fc.load_golden(n) is not None}
        if extra:
            out["also"] = extra
            out["ok"] = bool(out["ok"] and all(v["ok"] for v in extra.values()))
    finally:
        fc.close_model(mdl)
    return out


def cpu_baseline_other(config):
    """One loop iteration of BASELINE configs 1 / 2 / 5 on the host cores through the fp32 oracle (bounded sample, stated)."""
    from oracle.unet import SD15_CONFIG, SDXL_CONFIG, OracleUNet, random_state_dict
    from oracle.vae import SD_VAE_CONFIG, SDXL_VAE_CONFIG, OracleVAEDecoder, color_guidance_update, random_vae_state_dict
    threads_max = physical_cores()
    threads, cand, probe = pick_cpu_threads(threads_max)
    torch.set_num_threads(threads)
    xl = config == 5
    cfg, hw, F = (SDXL_CONFIG, 128, 7) if xl else (SD15_CONFIG, 64, 3 if config == 1 else 5)
    o = OracleUNet(cfg, random_state_dict(cfg, seed=1))
    g = torch.Generator().manual_seed(2)
    x, ctx = torch.randn(1, 4, hw, hw, generator=g), torch.randn(1, 77, cfg["cross_attention_dim"], generator=g)
    added = {"text_embeds": torch.randn(1, 1280, generator=g), "time_ids": torch.tensor([[8.0 * hw, 8.0 * hw, 0, 0, 8.0 * hw, 8.0 * hw]])} if xl else None
    with torch.no_grad():
        n_fwd = 1 if xl else F
        t0 = time.perf_counter()
        for _ in range(n_fwd):
            o.forward(x, 501.0, ctx, added)
        fwd_s = (time.perf_counter() - t0) / n_fwd
    step_s, sample = F * fwd_s, f"{n_fwd} batch-1 UNet forward(s) of the fp32 oracle ({fwd_s:.2f} s each) x {F} forwards per iteration"
    if config in (2, 5):
        vcfg = SDXL_VAE_CONFIG if xl else SD_VAE_CONFIG
        vh = 64                                                    # the guidance gradient is timed on a 64x64 latent (512x512 image) ...
        vae = OracleVAEDecoder(vcfg, random_vae_state_dict(vcfg, seed=3))
        lat, eps = torch.randn(1, 4, vh, vh, generator=g), torch.randn(1, 4, vh, vh, generator=g)
        masks = [(torch.rand(1, 1, 8 * vh, 8 * vh, generator=g) ** 2).repeat(1, 4, 1, 1) for _ in range(2)]
        rgb = [torch.rand(1, 3, 1, 1, generator=g)]
        t0 = time.perf_counter()
        color_guidance_update(vae, lat, eps, 0.37, vcfg["scaling_factor"], masks, rgb, 0.5, torch.rand(1, 4, vh, vh, generator=g))
        vs = time.perf_counter() - t0
        scale = (hw / vh) ** 2                                      # ... and scaled by the image area for the 1024x1024 case
        step_s += vs * scale
        sample += f" + 1 VAE decode + autograd gradient at {8 * vh}x{8 * vh} ({vs:.1f} s)" + (f" x {scale:.0f} (area) for {8 * hw}x{8 * hw}" if scale != 1 else "")
    return dict(value=1.0 / step_s, unit="steps/s", cores=threads, cores_physical=threads_max, kind="port", cpu=cpu_model_name(),
                cores_note=f"the fastest of {cand} on one ResNet block + one transformer block of the oracle's operators (seconds per probe: {probe})",
                sample=sample, step_seconds=step_s)


def cross_attention_block(dev, F=7):
    """The block BASELINE.json's north star names: SDXL cross-attention (attn2) = to_q GEMM -> attention over 77 keys (K/V from the
    per-prompt cache, font-size multipliers) -> to_out GEMM + bias + residual on the fp16 trunk, for the F batched streams of a step,
    as the ONE C-ABI call the engine's own path is made of (rt_op_cross_attn_block).  FLOPs = executed MFMA work (7.51 G
    shape A / 7.11 G shape B per stream)."""
    import ctypes as C
    from rich_text_to_image_amd.engine import _ptr, load_library
    lib = load_library()
    out = {}
    for name, N, Cc, H in (("A_4096x640", 4096, 640, 10), ("B_1024x1280", 1024, 1280, 20)):
        DP, M = 64, F * N
        g = torch.Generator(device=dev).manual_seed(1)
        bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(torch.bfloat16)
        x, wq, wo = bf(M, Cc), bf(H * DP, Cc, sc=Cc ** -0.5), bf(Cc, H * DP, sc=(H * DP) ** -0.5)
        K, VT = bf(5 * 96, H * DP), bf(H * DP, 5 * 96)
        wabs = torch.zeros(2, 96, device=dev); wabs[:, :77] = 1.0; wabs[1, 5:7] = 20.0
        wsgn = torch.ones(2, 96, device=dev)
        bo = torch.zeros(Cc, device=dev)
        trunk = torch.randn(M, Cc, generator=g, device=dev).to(torch.float16)
        q = torch.empty(M, H * DP, device=dev, dtype=torch.bfloat16); o = torch.empty_like(q); y = torch.empty_like(trunk)
        ia = lambda v: (C.c_int * F)(*v)
        prm, ws = ia([0, 4, 0, 4, 1, 2, 3][:F]), ia([-1, 1, -1, -1, -1, -1, -1][:F])      # as the engine: one font-size stream, the others plain (-1)

        def block():
            rc = lib.rt_op_cross_attn_block(_ptr(x), _ptr(wq), _ptr(wo), _ptr(bo), _ptr(K), _ptr(VT), 5 * 96, prm, ws, _ptr(wabs), _ptr(wsgn),
                                            _ptr(trunk), _ptr(y), _ptr(q), _ptr(o), F, N, Cc, H, DP, None)
            if rc != 0:
                raise RuntimeError(lib.rt_op_last_error().decode())
        def timed():
            for _ in range(3):
                block()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                block()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20
        flags = int(os.environ.get("RTDIFF_DEBUG_FLAGS", "0"))
        ms = timed()                                  # the engine's form for this shape
        lib.rt_op_gemm_debug(flags | 16 | 524288)     # same operator as to_q GEMM -> attn_kernel<CROSS> launch -> to_out GEMM (round 3's form)
        try:
            ms3 = timed()
        finally:
            lib.rt_op_gemm_debug(flags)
        probes = bool(lib.rt_op_probes_built())       # the fused / one-launch forms exist in `make PROBES=1` builds only (round 6)
        flops = F * (4.0 * N * Cc * H * DP + 4.0 * H * N * 77 * 64)
        out[name] = dict(ms=ms, tflops=flops / (ms * 1e-3) / 1e12, frac=flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         three_launch_ms=ms3, three_launch_frac=flops / (ms3 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS)
        if probes:
            lib.rt_op_gemm_debug(flags | 524288)      # round 4's form: to_q + attention fused (gemm16 EPI_XATTN) where the tiling allowed it
            try:
                ms4 = timed()
            finally:
                lib.rt_op_gemm_debug(flags)
            out[name].update(round4_form_ms=ms4, round4_form_frac=flops / (ms4 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS)
        if probes and Cc == 640:                                 # the one-launch register-chained form (xblock.hip; opt-in, NOT the engine's path: slower)
            lib.rt_op_gemm_debug(flags | 65536)
            try:
                ms1 = timed()
            finally:
                lib.rt_op_gemm_debug(flags)
            out[name].update(one_launch_xblock_ms=ms1, one_launch_xblock_frac=flops / (ms1 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS)
    out["note"] = ("rt_op_cross_attn_block: to_q + attention(77 keys, cached K/V, font-size softmax) + to_out(+bias, + fp16 trunk residual) "
                   "for the 7 streams of a step on a LayerNorm'd bf16 input, as three launches: to_q GEMM, cross77_kernel (cross77.hip: the 77-key "
                   "attention with K / V^T of one head in LDS, 64 or 128 queries per workgroup), to_out GEMM (in the engine the first and the "
                   "last additionally carry the folded LayerNorm / its partial sums, round 6); three_launch_* = to_q GEMM, the generic "
                   "attn_kernel<CROSS>, to_out GEMM (round 3); `make PROBES=1` builds add round4_form_* (to_q + attention as one kernel, "
                   "gemm16.hip EPI_XATTN) and one_launch_xblock_* (the whole block as one launch, xblock.hip): built, parity-tested, slower; "
                   "executed FLOPs")
    return out


def _config_parity(cfg):
    full = {1: "config1", 2: "config2_50", 5: "config5_50"}[cfg]
    short = {1: None, 2: "config2", 5: "config5"}[cfg]
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "fullschedule", full + ".pt")):
        full, short = short, None
    return pixel_parity(full, also=(short,) if short else ())


def other_config(args):
    """BASELINE.json configs 1, 2 and 5 (SURVEY 8d numbering) through the drop-in facade classes (tools/bench_configs.py holds the
    workloads): same JSON contract, N = 1 only.  `--steps K` = K scheduler steps of the named loop (PLMS runs K+1 iterations)."""
    assert args.gpus == 1, "configs 1/2/5 are single-GPU measurements"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc
    torch.cuda.set_device(0)
    bc.STEPS_OVERRIDE = args.steps
    bc.WARM_STEPS = max(1, args.warmup)
    r = {1: bc.config1, 2: bc.config2, 5: bc.config5}[args.config]()
    tf = r["tflop_per_iteration"] * r["value"]
    line = {"metric": "denoising steps/sec (" + r["workload"].split(",")[0] + ")", "value": r["value"], "unit": "steps/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / r["value"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": r["workload"], "baseline_config": args.config, "iterations_timed": r["iterations"]},
            "finite": r["finite"],
            **({"one_pass_guidance": r["one_pass_guidance"]} if "one_pass_guidance" in r else {}),
            "roofline": {"bound": "mfma", "kernel": "whole step (UNet forwards + VAE guidance where configured)", "achieved": tf,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS, "traffic": None},
            "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline_other(args.config),
            # the full schedule of the committed oracle trajectory of this configuration, latents and pixels (tests/test_fullschedule_gpu.py asserts the same)
            # round 6: configs 2 / 5 at BASELINE's own length / mask shape (51 PLMS iterations; 50-step schedule with 10 Voronoi cells, 30 iterations
            # across the blend), the short trajectories of round 5 under "also"
            "parity": None if args.no_cpu_baseline else _config_parity(args.config),
            "parity_tests": {1: "tests/test_fullsize_gpu.py::test_sd15_config1_rich_loop_matches_oracle (SD-v1.5 full architecture, PLMS loop vs the fp32 oracle)",
                             2: "tests/test_fullsize_gpu.py::test_sd15_full_architecture_stream_modes_match_oracle + test_full_width_vae_decode_and_guidance_gradient_match_oracle",
                             5: "tests/test_fullsize_gpu.py::test_sdxl_config3_rich_step_matches_oracle + test_full_width_vae_decode_and_guidance_gradient_match_oracle"}[args.config]}
    print(json.dumps(line), flush=True)


def dry_launch(args):
    """The N > 1 control path of this file without GPUs (tests/test_distributed_cpu.py): gloo rendezvous from the
    torch.distributed.run environment, ONE broadcast of an arena-shaped byte buffer from rank 0, barrier, max-over-ranks, rank 0
    prints the contract line with value = null."""
    from rich_text_to_image_amd import launcher
    rank, local_rank, world = launcher.init_distributed("gloo")
    if world != args.gpus:
        sys.exit(f"bench: --gpus {args.gpus} but the launch environment says WORLD_SIZE={world}")
    check = launcher.collective_self_check(nbytes=1 << 20)                  # the same self-check every real N > 1 launch starts with (gloo here)
    arena = (torch.arange(1 << 16, dtype=torch.int64) % 251).to(torch.uint8) if rank == 0 else torch.zeros(1 << 16, dtype=torch.uint8)
    t0 = time.perf_counter()
    calls = launcher.broadcast_tensor(arena, src=0)
    bcast_s = time.perf_counter() - t0
    ok = bool(torch.equal(arena, (torch.arange(1 << 16, dtype=torch.int64) % 251).to(torch.uint8)))
    launcher.barrier()
    dt = launcher.max_over_ranks(1.0 + rank)
    mine = launcher.shard_round_robin(list(range(2 * world)), rank, world)
    if rank == 0:
        print(json.dumps({"metric": "denoising steps/sec (SDXL 1024^2, 50-step, 4 regions)", "value": None, "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_launch": True, "scaling": "weak",
                          "weight_broadcast_calls": calls, "weight_broadcast_s": bcast_s, "arena_received": ok, "collective_check": check,
                          "max_over_ranks": dt, "requests_rank0": mine}), flush=True)
    if not ok:
        sys.exit("bench: dry launch: broadcast payload mismatch")
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50; 20 for --config 1, BASELINE's 20-step case)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--elide", action="store_true", help="skip reference forwards that cannot influence the output")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs of the N = 1 line (graph_replay, batched_2_requests, plain_pass, end_to_end: tools/end_to_end.py)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="warm up, then run ONLY the two event-profiled steps (one injected, one not) and print the line without the "
                         "timing / micro-benchmark legs: the command the rocprofv3 --pmc passes wrap (tools/pmc_passes.sh), so that the "
                         "counters and `roofline.flops_per_launch` describe the same launches")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise the multi-rank launch path only (rendezvous, one arena-sized broadcast, barrier, max-over-ranks) "
                         "on the gloo backend without touching a GPU, and print the JSON line with value = null (CPU test of --gpus N)")
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 5],
                    help="BASELINE.json configuration (SURVEY 8d numbering): 3 = the headline SDXL workload (default); "
                         "1 / 2 = SD-v1.5 (R=2 plain; R=4 + colour guidance); 5 = SDXL + colour guidance + background blend")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.config == 1 else 50
    if args.config != 3:
        return other_config(args)

    from rich_text_to_image_amd import launcher
    err = launcher.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus, require_gpus=not args.dry_launch)
    if err is not None:
        # not enough GPUs on this node: say so in the line the driver parses (no value) and on stderr
        print(f"bench: {err}", file=sys.stderr)
        print(json.dumps({"metric": "denoising steps/sec (SDXL 1024^2, 50-step, 4 regions)", "value": None, "unit": "steps/s",
                          "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": err}), flush=True)
        return
    if args.dry_launch:
        return dry_launch(args)
    from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
    # RTDIFF_DIST_BACKEND / RTDIFF_FORCE_DEVICE (tests only): N ranks on ONE GPU over gloo - RCCL cannot put two ranks on a device, and every
    # GPU lease of the build had one; the N > 1 control path of this file then runs end to end on real engines (tests/test_launcher_gpu.py)
    rank, local_rank, world = launcher.init_distributed(os.environ.get("RTDIFF_DIST_BACKEND"))
    if world != args.gpus:
        sys.exit(f"bench: --gpus {args.gpus} but the launch environment says WORLD_SIZE={world}")
    if "RTDIFF_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["RTDIFF_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    gloo = world > 1 and torch.distributed.get_backend() == "gloo"
    # N > 1: prove the collective path (64 MB pattern broadcast on the real backend, verified on every rank) before 5 GB of weights move
    collective_check = launcher.collective_self_check(device=torch.device("cpu") if gloo else None) if world > 1 else None
    if collective_check is not None and rank == 0:
        print(f"bench: collective self-check {collective_check}", file=sys.stderr, flush=True)

    R, hw, nsched, gs, isa, ibg = 4, 128, 50, 5.0, 0.5, 0.0
    eng = Engine(SDXL_CONFIG, hw, hw, device=local_rank, max_streams=8, max_prompts=8)

    # ---- weights: rank 0 draws + packs them, everyone else receives the packed bf16 arena (one broadcast)
    sd_cpu = None
    if rank == 0:
        keep_cpu = world == 1 and not args.no_cpu_baseline          # the CPU baseline is an N=1, rank-0 leg only
        g = torch.Generator(device=dev).manual_seed(0)
        sd_cpu = {} if keep_cpu else None
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
            if keep_cpu:
                sd_cpu[name] = t.cpu()
            eng.synchronize()
            del t
    bcast_s = launcher.broadcast_weights(eng, src=0)
    assert eng.weights_missing()[0] == 0

    # ---- per-rank independent request
    inp = synth_inputs(1000 + rank, R, hw, dev)
    ts, sig, init_sigma = euler_tables(nsched)
    eng.set_prompts(inp["emb"], inp["pooled"], inp["tid"])
    eng.set_masks(inp["masks"])
    eng.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
    lat0 = (inp["lat"] * init_sigma).to(dev)

    def reset():
        eng.set_schedule(0, ts, sig, nsched)
        eng.set_latents(lat0)

    def sched_index(i, k):
        # fewer timed steps than the schedule has: stride over it so that injected (t > 500) and non-injected steps are timed
        # in the schedule's own proportion; otherwise walk it cyclically
        return (i * nsched) // k if k < nsched else i % nsched

    def run(k):
        for i in range(k):
            eng.region_step(sched_index(i, k), gs, isa, ibg, xl=True, elide=args.elide)

    reset()
    run(max(0, args.warmup - 1))
    if args.warmup > 0:          # also warm (and tile-tune) the shapes of the non-injected half of the schedule
        eng.region_step(nsched - 1, gs, isa, ibg, xl=True, elide=args.elide)
    eng.synchronize()
    reset()
    launcher.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(0 if args.roofline_only else args.steps)
    eng.synchronize()
    torch.cuda.synchronize()
    dt_local = max(time.perf_counter() - t0, 1e-9)
    launcher.barrier()
    dt = launcher.max_over_ranks(dt_local, device=dev if (world > 1 and not gloo) else "cpu")
    final = eng.read_latents(hw, hw)
    finite = bool(torch.isfinite(final).all())

    # ---- roofline leg (outside the timed region): HIP events around every MFMA kernel launch on the engine
    # stream for one injected step (i=0) and one non-injected step (i=nsched-1)
    roof = None
    prof = None
    step_flops = None
    if rank == 0:
        reset()
        per_step = []
        for i_prof in (0, nsched - 1):                       # one injected (t > 500) and one non-injected step
            eng.profile_enable(True)
            eng.region_step(i_prof, gs, isa, ibg, xl=True, elide=False)
            per_step.append(eng.profile_read())
            eng.profile_enable(False)
        prof = {k: {f: per_step[0][k][f] + per_step[1][k][f] for f in ("launches", "total_ms", "total_flops")} for k in per_step[0]}
        # MFMA FLOPs the engine actually executed per step (GEMM + convolution + attention launches; the region streams of an
        # injected step skip their Q|K projections): weights the whole-step rate by the timed schedule's own mix of the two kinds
        f_inj, f_non = (sum(v["total_flops"] for v in ps.values()) for ps in per_step)
        idx = [sched_index(i, args.steps) for i in range(args.steps)]
        n_inj = sum(1 for i in idx if ts[i] > (1.0 - isa) * 1000.0)
        step_flops = dict(injected=f_inj, non_injected=f_non, timed_injected_steps=n_inj, timed_steps=len(idx),
                          mean=(n_inj * f_inj + (len(idx) - n_inj) * f_non) / max(1, len(idx)))
        dom = max(prof, key=lambda k: prof[k]["total_ms"])
        p = prof[dom]
        achieved = p["total_flops"] / (p["total_ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch of the dominant kernel class come from separate rocprofv3 --pmc passes over
        # `bench.py --roofline-only` (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process); tools/pmc_traffic.py keeps
        # the dispatches of the two profiled steps only, i.e. the launches `flops_per_launch` is averaged over
        traffic = None
        tfile = None
        try:
            tfile = next(f for f in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
            cls = {"gemm_kernel<A_DENSE>": "gemm_dense", "gemm_kernel<A_CONV3*>": "gemm_conv", "attn_kernel<self>": "attn_self",
                   "attn_kernel<cross>": "attn_cross", "gemm16_kernel<EPI_XATTN> (to_q + cross-attention)": "xattn_fused"}[dom]
            traffic = tj["classes"][cls]["hbm_bytes_per_launch"]
            traffic_us = tj["classes"][cls].get("avg_us")          # the class's average launch duration IN THE COUNTER PASSES (counters perturb: not avg_launch_us below)
        except Exception:
            traffic_us = None
        roof = dict(bound="mfma", kernel=dom + " (dense GEMM launches: gemm16_kernel<A_DENSE> / gemm_kernel<A_DENSE>)" if dom == "gemm_kernel<A_DENSE>" else dom,
                    achieved=achieved, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                    frac=achieved / PEAK_BF16_TFLOPS, traffic=traffic, traffic_source=f"profiles/{tfile} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --roofline-only`, the dispatches of the two profiled steps; gfx950 x2 FETCH correction)" if traffic else None,
                    traffic_avg_launch_us=traffic_us if traffic else None,
                    launches=p["launches"],
                    avg_launch_us=p["total_ms"] * 1e3 / max(1, p["launches"]),
                    flops_per_launch=p["total_flops"] / max(1, p["launches"]),
                    per_kernel={k: dict(launches=v["launches"], total_ms=round(v["total_ms"], 3),
                                        tflops=(v["total_flops"] / (v["total_ms"] * 1e-3) / 1e12) if v["total_ms"] > 0 else 0.0)
                                for k, v in prof.items()})

    # ---- informational second timing: forwards whose results nothing consumes are skipped (after the injection window and
    # the background-blend step the uncond_ref / text_ref forwards are dead in the reference too).  Same
    # final latents; NOT the headline value (the reference runs those forwards).
    elided = None
    if not args.elide and not args.roofline_only:
        reset()
        for i in range(min(2, nsched)):
            eng.region_step(i, gs, isa, ibg, xl=True, elide=True)
        eng.region_step(nsched - 1, gs, isa, ibg, xl=True, elide=True)
        eng.synchronize(); reset()
        launcher.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            eng.region_step(sched_index(i, args.steps), gs, isa, ibg, xl=True, elide=True)
        eng.synchronize(); torch.cuda.synchronize()
        dt_e = launcher.max_over_ranks(time.perf_counter() - t0, device=dev if (world > 1 and not gloo) else "cpu")
        same = bool(torch.allclose(eng.read_latents(hw, hw), final, rtol=0, atol=0)) if args.steps <= nsched else None
        elided = dict(value=world * args.steps / dt_e, ms_per_step=dt_e / args.steps * 1e3, identical_latents=same)

    xblock = None
    if rank == 0 and not args.roofline_only:
        try:
            xblock = cross_attention_block(dev)
        except Exception as ex:          # the headline line must still print
            xblock = {"error": repr(ex)}

    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline and not args.roofline_only and sd_cpu is not None:
        threads = physical_cores()
        # one engine stream on the oracle's inputs: prompt 0 (negative), t = 801, unscaled latents
        gp = torch.Generator().manual_seed(4242)
        px = torch.randn(1, 4, hw, hw, generator=gp)
        eng.set_fontsize(None, None)
        got = eng.unet_forward(px.to(dev), 801.0, [0]).cpu()
        cpu, ref = cpu_baseline(sd_cpu, threads, px, inp["emb"][:1].cpu(), inp["pooled"][:1].cpu(), inp["tid"], 801.0,
                                full_step=True, all_ctx=inp["emb"].cpu(), all_pooled=inp["pooled"].cpu())
        rel = float(((got - ref).pow(2).sum() / ref.pow(2).sum()).sqrt())
        parity = dict(rel_l2=rel, tol=PARITY_TOL, ok=bool(rel <= PARITY_TOL),
                      config="SDXL-base full architecture, 1 UNet forward (latent 128x128, t=801, negative-prompt stream) vs the fp32 CPU oracle")

    # ---- the full schedule against the committed oracle trajectory, in latents and in pixels (config 3: all 50 Euler steps)
    pixels = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.roofline_only:
        try:
            # the benched workload's own 50-step schedule (+ the 10-step one that crosses the injection boundary between two checkpoints)
            pixels = pixel_parity("config3_50", also=("config3", "config3_unit")) if os.path.exists(os.path.join(ROOT, "tests", "golden", "fullschedule", "config3_50.pt")) else pixel_parity("config3")
        except Exception as ex:          # the headline line must still print
            pixels = {"error": repr(ex)}

    # ---- the other half of an image and the secondary modes (N = 1, rank 0, outside every timed region; tools/end_to_end.py).
    # They re-program the engine (prompts, schedule, capture), so they come after everything that reads the headline state.
    extras = {}
    if rank == 0 and world == 1 and not args.roofline_only and not args.no_extras:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import end_to_end as e2e

        def graph_replay():
            # the same `steps` steps captured into one HIP graph each and replayed: does the host side of ~1500 launches per step cost anything?
            side = torch.cuda.Stream()
            eng.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))      # (the parity leg cleared the multipliers)
            eng.synchronize(); eng.set_stream(side.cuda_stream)
            try:
                reset(); torch.cuda.synchronize()
                graphs = []
                for i in range(args.steps):
                    gph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gph, stream=side):
                        eng.region_step(sched_index(i, args.steps), gs, isa, ibg, xl=True, elide=args.elide)
                    graphs.append(gph)
                reset(); torch.cuda.synchronize()
                for gph in graphs[:2]:
                    gph.replay()
                torch.cuda.synchronize(); reset(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for gph in graphs:
                    gph.replay()
                torch.cuda.synchronize()
                dtg = time.perf_counter() - t0
                same = bool(torch.equal(eng.read_latents(hw, hw), final))
            finally:
                torch.cuda.synchronize(); eng.set_stream(None)
            return dict(ms_per_step=dtg / args.steps * 1e3, value=args.steps / dtg, identical_latents=same,
                        note="each timed step captured into its own HIP graph (the step index is a launch argument) and replayed back to back")
        def split_estimate():
            # intra-image split over 2 GPUs (launcher.split_region_step, `sample.py --gpus 2 --split_image`): the forwards of the two stream
            # ranges timed one after the other on THIS GPU - the step of a 2-GPU run costs the longer one + one 1.8 MB exchange + the
            # epilogue.  An estimate from one device, labelled as such: the leases here have one GPU.
            reset(); eng.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
            out = {}
            for label, idx in (("injected_step", 0), ("plain_step", nsched - 1)):
                parts = []
                for part in range(2):
                    for _ in range(2):
                        rng = eng.region_step_part(idx, gs, isa, ibg, True, part, 2)
                    eng.synchronize(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        eng.region_step_part(idx, gs, isa, ibg, True, part, 2)
                    eng.synchronize(); torch.cuda.synchronize()
                    parts.append(dict(streams=[rng[0], rng[0] + rng[1] - 1], ms=(time.perf_counter() - t0) / 5 * 1e3))
                out[label] = dict(parts=parts, longer_part_ms=max(p_["ms"] for p_ in parts))
            out["predicted_ms_per_step_2_gpus"] = 0.5 * (out["injected_step"]["longer_part_ms"] + out["plain_step"]["longer_part_ms"])
            out["note"] = ("forwards of the two stream ranges of launcher.split_region_step ({uncond, base, uncond_ref} | {text_ref, regions} while "
                           "injecting, 4 | 3 streams otherwise) timed on ONE GPU; a 2-GPU step = the longer range + one exchange of the noise "
                           "predictions (1.8 MB) + the epilogue; ESTIMATE - not a 2-GPU measurement")
            reset()
            return out
        legs = [("intra_image_split_estimate", split_estimate), ("graph_replay", graph_replay),
                ("batched_2_requests", lambda: e2e.two_requests(eng, lambda sd_: synth_inputs(sd_, R, hw, dev), hw, nsched, args.steps, gs, isa, sched_index, ts, sig, init_sigma)),
                ("plain_pass", lambda: e2e.plain_pass(eng, inp, hw)),
                ("end_to_end", lambda: e2e.end_to_end(eng, hw)),
                # informational: the same image with the colour-guidance pass on a one-pass bf16 VAE engine (opt-in, sample.py --guidance_precision bf16)
                ("end_to_end_one_pass_guidance", lambda: e2e.end_to_end(eng, hw, one_pass_guidance=True))]
        for name, fn in legs:
            try:
                extras[name] = fn()
            except Exception as ex:          # the headline line must still print
                extras[name] = {"error": repr(ex)}

    if rank == 0:
        value = None if args.roofline_only else world * args.steps / dt
        # executed MFMA FLOPs per step (profile leg) rather than 7 x the nominal forward: injected steps skip the region streams' Q|K GEMMs
        step_tflop = step_flops["mean"] / 1e12 if step_flops else 7 * SDXL_FWD_GFLOP / 1e3
        line = {
            "metric": "denoising steps/sec (SDXL 1024^2, 50-step, 4 regions)",
            "value": value, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None if args.roofline_only else dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "SDXL RegionDiffusionXL 1024x1024, R=4 regions, inject_selfattn=0.5, 50-step Euler, "
                                   "CFG 5.0, 7 UNet forwards/step batched, random-init SDXL-base weights",
                       "global_batch": world, "parallelism": f"seed-parallel x{world} (1 weight broadcast, no per-step collectives)",
                       "elide_dead_forwards": bool(args.elide)},
            "whole_step_tflops_per_gpu": None if args.roofline_only else step_tflop / (dt / args.steps),
            "whole_step_mfma_frac": None if args.roofline_only else step_tflop / (dt / args.steps) / PEAK_BF16_TFLOPS,
            "executed_flops_per_step": step_flops, "nominal_tflop_per_step": 7 * SDXL_FWD_GFLOP / 1e3,
            "weight_broadcast_s": bcast_s, "weight_broadcast_calls": launcher.LAST_BROADCAST_CALLS, "collective_check": collective_check, "finite": finite,
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "pixel_parity": pixels, "cross_attention_block": xblock, "elide_dead_forwards_timing": elided,
            "timed_schedule_indices": [sched_index(i, args.steps) for i in range(min(args.steps, nsched))],
        }
        line.update(extras)
        print(json.dumps(line), flush=True)
        if parity is not None and not parity["ok"]:
            sys.exit(f"bench: full-architecture parity FAILED: rel-L2 {parity['rel_l2']:.3e} > {PARITY_TOL}")
        if pixels is not None and pixels.get("ok") is False:
            sys.exit(f"bench: full-schedule pixel parity FAILED: {pixels}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
