"""The other half of an image: measurements that `bench.py` adds to its JSON line at N = 1.

  plain_pass(eng, ...)      steps/s of the plain-text pass the reference runs before every rich-text pass
                            (models/region_diffusion_sdxl.py:879-914: batch-2 CFG forward per step, `sample.py:59-75`) WITH the
                            token-map capture on (xl.py:959-1016: the hooks accumulate head-averaged maps after the 10th call), and
                            the HBM roofline of `attn_store_kernel` (csrc/attn_store.hip) from HIP events around its launches.
  end_to_end(eng, ...)      wall clock of `sample.generate`'s stages - plain pass, get_token_maps x 2 (utils/attention_utils.py:233-341,
                            SpectralClustering(n_init=100) on the host), rich pass, VAE decode - the timings the reference prints
                            (sample.py:59,75,96,113).  Full SDXL architecture, random-init weights, synthetic tokenizer / text
                            embeddings (no CLIP vocabulary or weights offline: the text encoders are NOT in these timings).
  two_requests(eng, ...)    two independent requests on one GPU (two engines, one HIP stream each): aggregate steps/s and per-image
                            latency, reported beside the one-request headline, never instead of it.
"""
import math
import time
import types

import torch

HBM_PEAK_TBS = 8.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 measured by a float4 copy)
HBM_MEASURED_TBS = 6.29


def euler_tables(n):
    import numpy as np
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, 0)
    train_sig = (((1 - ac) / ac) ** 0.5).numpy().astype("float64")
    ts = (np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype("float32") + 1
    sig = np.concatenate([np.interp(ts, np.arange(0, 1000), train_sig), [0.0]]).astype("float32")
    return ts.tolist(), sig.tolist(), float((sig.max() ** 2 + 1) ** 0.5)


def recorded_modules(eng):
    """The modules the SDXL token-map hooks keep (region_diffusion_sdxl.RegionDiffusionXL._store_begin)."""
    from rich_text_to_image_amd.attention_utils import CrossAttentionLayers_XL
    out = []
    for name, max_tokens, _ in eng.attn_modules():
        if (name.endswith("attn1") and max_tokens <= 1024) or name in CrossAttentionLayers_XL:
            out.append(name)
    return out


def plain_pass(eng, inp, hw, steps=41, gs=5.0):
    """`steps` plain steps as sample.py's first pass runs them (default --sample_steps 41), capture enabled on the recorded modules;
    the maps start accumulating at the 11th call of a module (xl.py:977,988), exactly as in the reference."""
    dev = inp["emb"].device
    ts, sig, init_sigma = euler_tables(steps)
    eng.set_prompts(inp["emb"][[0, -1]], inp["pooled"][[0, -1]], inp["tid"])          # [negative, base]
    rec = recorded_modules(eng)
    names = [n for n, _, _ in eng.attn_modules()]
    for n in names:
        eng.attn_store_enable(n, 1 if n in rec else 0)
    lat0 = (inp["lat"] * init_sigma).to(dev)

    def run(k, capture=True):
        eng.set_schedule(0, ts, sig, steps)
        eng.set_latents(lat0)
        eng.attn_store_reset()
        for i in range(k):
            eng.plain_step(i, gs)
        eng.synchronize()
    run(3)                                                           # warm the batch-2 shapes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same pass without capture (what the maps cost)
    for n in names:
        eng.attn_store_enable(n, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt_off = time.perf_counter() - t0
    # HIP events around the store launches of two capturing steps (calls 11, 12 of every module)
    for n in names:
        eng.attn_store_enable(n, 1 if n in rec else 0)
    eng.set_schedule(0, ts, sig, steps); eng.set_latents(lat0); eng.attn_store_reset()
    for i in range(10):
        eng.plain_step(i, gs)
    eng.profile_enable(True)
    eng.plain_step(10, gs); eng.plain_step(11, gs)
    st = eng.profile_read_store()
    eng.profile_enable(False)
    for n in names:
        eng.attn_store_enable(n, 0)
    eng.attn_store_reset()
    tbs = st["total_bytes"] / max(st["total_ms"] * 1e-3, 1e-12) / 1e12
    return dict(steps=steps, steps_per_s=steps / dt, ms_per_step=dt / steps * 1e3, ms_per_step_without_capture=dt_off / steps * 1e3,
                capture_cost_ms_per_capturing_step=(dt - dt_off) / max(1, steps - 10) * 1e3,
                workload=f"SDXL 1024^2 plain pass: {steps}-step Euler, CFG {gs}, batch-2 forward per step, token-map capture on "
                         f"{len(rec)} modules from the 11th call (xl.py:977,988)",
                recorded_modules=len(rec),
                attn_store_roofline=dict(bound="hbm", kernel="attn_store_kernel", launches=st["launches"],
                                         avg_launch_us=st["total_ms"] * 1e3 / max(1, st["launches"]),
                                         bytes_per_launch=st["total_bytes"] / max(1, st["launches"]),
                                         achieved=tbs, peak=HBM_PEAK_TBS, unit="TB/s", frac=tbs / HBM_PEAK_TBS,
                                         frac_of_measured_copy_rate=tbs / HBM_MEASURED_TBS,
                                         mfma_tflops=st["total_flops"] / max(st["total_ms"] * 1e-3, 1e-12) / 1e12,
                                         note="algorithmic bytes: fp32 accumulator read + written once, Q / K rows of the recorded stream read once; "
                                              "HIP events on the engine stream around the launches of two capturing steps"))


class _WordTokenizer:
    """Synthetic stand-in for CLIPTokenizer (no vocabulary offline): what richtext_utils needs is `_tokenize(str) -> ['word</w>', ...]`."""
    model_max_length = 77

    def _tokenize(self, text):
        import re
        return [w.lower() + "</w>" for w in re.findall(r"[A-Za-z0-9]+|[^\sA-Za-z0-9]", text)]


def _synthetic_text_encoders(dev):
    """prompt strings -> deterministic pseudo-embeddings of the SDXL shapes ([n, 77, 2048], [n, 1280]); NOT a text encoder."""
    def embed(texts):
        e, p = [], []
        for t in texts:
            g = torch.Generator().manual_seed(sum((i + 1) * ord(c) for i, c in enumerate(t)) % (2 ** 31))
            e.append(torch.randn(77, 2048, generator=g)); p.append(torch.randn(1280, generator=g))
        return torch.stack(e).to(dev), torch.stack(p).to(dev)

    def call(prompt, negative_prompt):
        pe, pp = embed([prompt] if isinstance(prompt, str) else list(prompt))
        ne, npool = embed([negative_prompt] if isinstance(negative_prompt, str) else list(negative_prompt or [""]))
        return pe, ne[:1], pp, npool[:1]
    return call


RICH_TEXT = {"ops": [{"insert": "a "}, {"attributes": {"font": "slabo"}, "insert": "night sky"}, {"insert": " filled with stars above a "},
                     {"attributes": {"color": "#ff0000", "size": "60px"}, "insert": "barn"}, {"insert": " next to a "},
                     {"attributes": {"link": "a wooden fence covered in snow"}, "insert": "fence"}, {"insert": "\n"}]}


def end_to_end(eng, hw=128, steps=41, seed=6, inject_selfattn=0.5, num_segments=9, use_guidance=True, one_pass_guidance=False):
    """sample.generate on the full SDXL architecture (the bench engine is handed to the facade: same weights, no second arena)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_configs import random_vae
    from rich_text_to_image_amd import attention_utils, sample
    from rich_text_to_image_amd.engine import SDXL_CONFIG, SDXL_VAE_CONFIG
    from rich_text_to_image_amd.region_diffusion_sdxl import RegionDiffusionXL
    dev = torch.device(f"cuda:{eng.device}")
    t0 = time.perf_counter()
    vae = random_vae(SDXL_VAE_CONFIG, hw, hw, precise=True)
    t_vae_build = time.perf_counter() - t0
    m = RegionDiffusionXL(device=eng.device, unet_state_dict="random0", config=SDXL_CONFIG, vae=vae, vae_scaling_factor=SDXL_VAE_CONFIG["scaling_factor"],
                          tokenizer=_WordTokenizer(), text_encoders=_synthetic_text_encoders(dev))
    m.unet._engines[(hw, hw)] = eng
    if one_pass_guidance:                                            # opt-in: the guidance pass alone on a one-pass bf16 engine (LABNOTES R6.6)
        m.guidance_vae = random_vae(SDXL_VAE_CONFIG, hw, hw, precise=False)
    param = {"text_input": RICH_TEXT, "height": 8 * hw, "width": 8 * hw, "guidance_weight": 5.0, "steps": steps, "noise_index": seed, "negative_prompt": ""}
    # time the clustering inside get_token_maps separately (is it worth a GPU kernel? VERDICT r3 item 3)
    import sklearn.cluster as skc
    spent = {"spectral_s": 0.0, "calls": 0}
    orig = skc.SpectralClustering.fit_predict

    def timed_fit(self, X, y=None):
        t = time.perf_counter()
        try:
            return orig(self, X, y)
        finally:
            spent["spectral_s"] += time.perf_counter() - t; spent["calls"] += 1
    skc.SpectralClustering.fit_predict = timed_fit
    try:
        lat = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(seed))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plain, rich, t = sample.generate(m, param, "SDXL", None, color_guidance_weight=0.5 if use_guidance else 0.0, inject_selfattn=inject_selfattn,
                                         num_segments=num_segments, inject_background=0.0, latents=lat.clone())
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    finally:
        skc.SpectralClustering.fit_predict = orig
    # the final decode alone (it is inside both `plain` and `rich`: sample() returns PIL images like the reference)
    z = torch.randn(1, 4, hw, hw, device=dev)
    vae.decode(z); torch.cuda.synchronize()
    t0 = time.perf_counter(); vae.decode(z); torch.cuda.synchronize()
    t_dec = time.perf_counter() - t0
    import numpy as np
    fin = bool(np.isfinite(np.asarray(rich.images[0], dtype=np.float32)).all())
    n_regions = len(m.masks) if m.masks is not None else None
    m.unet._engines = {}                                             # the engine belongs to the caller
    if m.guidance_vae is not None:
        m.guidance_vae.close(); m.guidance_vae = None
    return dict(seconds_total=total, one_pass_guidance=bool(one_pass_guidance), plain_pass_s=t["plain"], token_maps_x2_s=t["token_maps"], rich_pass_s=t["rich"],
                spectral_clustering_s=spent["spectral_s"], spectral_clustering_calls=spent["calls"], vae_decode_s=t_dec,
                steps=steps, regions=n_regions, finite=fin,
                workload=f"sample.generate (sample.py:56-113): SDXL 1024^2, {steps} steps, CFG 5.0, inject_selfattn={inject_selfattn}, "
                         f"num_segments={num_segments}, colour guidance {'on (precise VAE)' if use_guidance else 'off'}; plain and rich timings "
                         "include their final precise-VAE decode and the uint8 / PIL hand-off",
                not_included="CLIP tokenizer / text encoders (no vocabulary or weights offline: synthetic embeddings), model construction "
                             f"(precise VAE build {t_vae_build:.2f} s)")


def two_requests(eng, make_inputs, hw, nsched, steps, gs, isa, sched_index, ts, sig, init_sigma):
    """Two independent rich-text requests on ONE GPU: two engines (the second arena is a device-to-device copy of the first), each
    on its own HIP stream, stepped alternately by one host thread - the launches of the two requests interleave on the chip, so the
    prologue / epilogue / kernel boundary of one request's launch overlaps the main loop of the other's.  Reports aggregate steps/s
    AND per-image latency; a secondary number, the headline stays one request per GPU."""
    from rich_text_to_image_amd import launcher
    from rich_text_to_image_amd.engine import SDXL_CONFIG, Engine
    dev = f"cuda:{eng.device}"
    eng2 = Engine(SDXL_CONFIG, hw, hw, device=eng.device, max_streams=8, max_prompts=8)
    eng.synchronize()
    launcher.arena_tensor(eng2).copy_(launcher.arena_tensor(eng))
    torch.cuda.synchronize()
    eng2.arena_mark_bound()
    engines = (eng, eng2)
    lats = []
    for k, e in enumerate(engines):
        inp = make_inputs(2000 + k)
        e.set_prompts(inp["emb"], inp["pooled"], inp["tid"])
        e.set_masks(inp["masks"])
        e.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
        lats.append((inp["lat"] * init_sigma).to(dev))

    def reset():
        for e, l in zip(engines, lats):
            e.set_schedule(0, ts, sig, nsched)
            e.set_latents(l)

    def run(k):
        for i in range(k):
            for e in engines:
                e.region_step(sched_index(i, k), gs, isa, 0.0, xl=True, elide=False)
        for e in engines:
            e.synchronize()
    reset(); run(2)
    for e in engines:
        e.region_step(nsched - 1, gs, isa, 0.0, xl=True, elide=False)
    for e in engines:
        e.synchronize()
    reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fin = all(bool(torch.isfinite(e.read_latents(hw, hw)).all()) for e in engines)
    eng2.close()
    return dict(requests=2, aggregate_steps_per_s=2 * steps / dt, per_image_ms_per_step=dt / steps * 1e3, steps=steps, finite=fin,
                how="two engines (second weight arena = device copy of the first), one HIP stream each, stepped alternately from one host thread; "
                    "same workload per request as the headline line")
