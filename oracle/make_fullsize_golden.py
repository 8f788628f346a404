"""TEST INFRASTRUCTURE: full-SCHEDULE trajectories of the fp32 CPU oracle at the architectures BASELINE.json names.

    python -m oracle.make_fullsize_golden config1 config2 config3 config5 [config3_50]      # in the build container (CPU time is free there)

writes tests/golden/fullschedule/<case>.pt = latents at a list of loop iterations + the final image (uint8, decoded by the
oracle VAE as the reference decodes it: rd.py:227-236,267-271 / xl.py:916-944).  tests/test_fullschedule_gpu.py runs the HIP
engine over the SAME schedules and compares per checkpoint and in pixels.

What is followed (file:line under /root/reference):
  * config 1 = RegionDiffusion.produce_latents, models/region_diffusion.py:86-174: SD-v1.5 @ 64x64 latent, R = 2, 20 requested
    steps = 21 PLMS iterations, CFG 8.5, one font-size token
  * config 3 = RegionDiffusionXL.sample rich branch, models/region_diffusion_sdxl.py:779-878: SDXL-base @ 128x128, R = 4,
    inject_selfattn = 0.5, CFG 5, two font-size tokens; 10 Euler steps (t > 500 for iterations 0..4: the injection boundary
    falls between checkpoints 5 and 6) and, as `config3_50`, the full 50 steps
  * config 5 = the same loop with use_guidance (xl.py:849-867, precise fp32 SDXL VAE), inject_background = 0.5, CFG 7.5,
    4 Euler steps @ 128x128 (blend at iteration 2, xl.py:870)

Nothing large is committed: the UNet / VAE weights are `oracle.unet.random_state_dict(cfg, seed)` /
`oracle.vae.random_vae_state_dict(cfg, seed)` - drawn by torch's CPU generator from a seed, regenerated bit-identically on the
GPU box (same image, same torch) and checked there against the fingerprint stored in the file.  The inputs are functions of a
seed too (`case_inputs`).  The loops are oracle.region_loop's, pinned against the unmodified reference loops by
tests/test_oracle_vs_reference.py.
"""
import os
import sys
import time

import torch

from .region_loop import rich_loop_sd, rich_loop_xl
from .schedulers import OracleEuler, OraclePNDM
from .unet import SD15_CONFIG, SDXL_CONFIG, OracleUNet, random_state_dict
from .vae import SD_VAE_CONFIG, SDXL_VAE_CONFIG, OracleVAEDecoder, random_vae_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "fullschedule")

CASES = {
    # name: model, latent size, regions, requested steps, CFG, inject_selfattn, inject_background, guided, checkpoints (1-based loop iterations)
    "config1": dict(model="sd15", hw=64, R=2, steps=20, gs=8.5, isa=0.0, ibg=0.0, guided=False, unet_seed=101, vae_seed=201, seed=301,
                    word_pos=[2], font_size=[3.0], checkpoints=[1, 2, 3, 6, 11, 16, 21]),
    "config2": dict(model="sd15", hw=64, R=4, steps=10, gs=7.5, isa=0.0, ibg=0.0, guided=True, unet_seed=101, vae_seed=201, seed=302,
                    word_pos=[2], font_size=[3.0], n_color=2, color_weight=20.0, checkpoints=[1, 2, 3, 6, 11]),
    "config3": dict(model="sdxl", hw=128, R=4, steps=10, gs=5.0, isa=0.5, ibg=0.0, guided=False, unet_seed=103, vae_seed=203, seed=303,
                    word_pos=[5, 6], font_size=[20.0, 20.0], checkpoints=[1, 2, 5, 6, 10]),
    "config3_50": dict(model="sdxl", hw=128, R=4, steps=50, gs=5.0, isa=0.5, ibg=0.0, guided=False, unet_seed=103, vae_seed=203, seed=303,
                       word_pos=[5, 6], font_size=[20.0, 20.0], checkpoints=[1, 2, 5, 10, 15, 20, 25, 26, 30, 35, 40, 45, 50]),
    "config5": dict(model="sdxl", hw=128, R=4, steps=4, gs=7.5, isa=0.0, ibg=0.5, guided=True, unet_seed=103, vae_seed=203, seed=305,
                    word_pos=[4], font_size=[8.0], n_color=1, color_weight=20.0, checkpoints=[1, 2, 3, 4]),
    # round 6: the cases at BASELINE's own shape, and a regime in which the accumulated update dominates the start noise
    # config3_unit: config3_50 started from UNIT-variance latents (`latents` handed over at 1 / init_noise_sigma, so prepare_latents'
    #   multiplication, xl.py:536, gives std 1 instead of 14.6): sum_k (sigma_k+1 - sigma_k) eps_k is then an order of magnitude above lat0
    #   and a relative error of the latents IS a relative error of the update.
    "config3_unit": dict(model="sdxl", hw=128, R=4, steps=50, gs=5.0, isa=0.5, ibg=0.0, guided=False, unet_seed=103, vae_seed=203, seed=303,
                         word_pos=[5, 6], font_size=[20.0, 20.0], unit_start=True, checkpoints=[1, 2, 5, 10, 15, 20, 25, 26, 30, 35, 40, 45, 50]),
    # config2_50: BASELINE config 2 at its own length - 50 requested steps = 51 PLMS iterations, 4 regions, colour guidance on 2 regions
    #   (weight 20 as in `config2`, not BASELINE's 1: with seeded random weights the gradient is small, and a guidance term below the bf16 noise would test nothing)
    "config2_50": dict(model="sd15", hw=64, R=4, steps=50, gs=7.5, isa=0.0, ibg=0.0, guided=True, unet_seed=101, vae_seed=201, seed=302,
                       word_pos=[2], font_size=[3.0], n_color=2, color_weight=20.0, checkpoints=[1, 2, 3, 6, 11, 21, 31, 41, 51]),
    # config5_50: BASELINE config 5 on its 50-step schedule, "10 segments" = 10 Voronoi cells on 32 x 32 dealt to the 4 regions,
    #   footnote (font size) + colour guidance, inject_background 0.5 => blend after loop index 25 (xl.py:870); the loop is stopped after
    #   iteration 30 (`stop_after`: 8 forwards + one fp32 VAE forward / backward per step cost ~3 CPU-minutes each in the build container)
    "config5_50": dict(model="sdxl", hw=128, R=4, steps=50, gs=7.5, isa=0.0, ibg=0.5, guided=True, unet_seed=103, vae_seed=203, seed=305,
                       word_pos=[4], font_size=[8.0], n_color=1, color_weight=20.0, masks="voronoi10", stop_after=30,
                       checkpoints=[1, 2, 5, 10, 15, 20, 25, 26, 27, 30]),
}


class StopLoop(Exception):
    """Raised by the trace / callback of a `stop_after` case to leave the loop after the last recorded iteration."""


def smooth_masks(R, hw, g):
    """R soft region masks that sum to 1 (attention_utils.py:325-329 normalises the same way), 4 identical channels."""
    m = torch.softmax(torch.randn(R, 1, hw // 4, hw // 4, generator=g) * 4, dim=0)
    m = torch.nn.functional.interpolate(m, size=(hw, hw), mode="bilinear", align_corners=False)
    return (m / (m.sum(0, keepdim=True) + 1e-8)).repeat(1, 4, 1, 1)


def voronoi_masks(R, hw, g, cells=10, grid=32):
    """`num_segments = 10` of BASELINE config 5: 10 Voronoi cells on the 32 x 32 self-attention grid, every region owning
    at least one, then exactly what get_token_maps does with its binary cluster maps (attention_utils.py:322-327): bicubic antialiased
    resize to the latent size, clamp(0, 1), division by the sum over regions + 1e-8.  4 identical channels; last mask = background."""
    pts = torch.rand(cells, 2, generator=g) * grid
    owner = torch.cat([torch.arange(R), torch.randint(0, R, (cells - R,), generator=g)])[torch.randperm(cells, generator=g)]
    yy, xx = torch.meshgrid(torch.arange(grid) + 0.5, torch.arange(grid) + 0.5, indexing="ij")
    d = (yy[None] - pts[:, 0, None, None]) ** 2 + (xx[None] - pts[:, 1, None, None]) ** 2
    region = owner[d.argmin(0)]                                                   # [grid, grid] region index per cell of the grid
    maps = torch.stack([(region == r).to(torch.float64) for r in range(R)])      # binary, float64 like the numpy maps of the reference
    assert all(m.sum() > 0 for m in maps)
    m = torch.cat([torch.nn.functional.interpolate(t[None, None], (hw, hw), mode="bicubic", antialias=True)[0] for t in maps]).clamp(0, 1)
    m = m / (m.sum(0, True) + 1e-8)
    return m.float().unsqueeze(1).repeat(1, 4, 1, 1)


def case_inputs(name):
    """Every input of a case from its seed (CPU generator): identical here and on the GPU box."""
    c = CASES[name]
    xl = c["model"] == "sdxl"
    hw, R = c["hw"], c["R"]
    g = torch.Generator().manual_seed(c["seed"])
    d = {"emb": torch.randn(R + 1, 77, 2048 if xl else 768, generator=g)}
    if xl:
        d["pooled"] = torch.randn(R + 1, 1280, generator=g)
        d["time_ids"] = torch.tensor([[8.0 * hw, 8.0 * hw, 0, 0, 8.0 * hw, 8.0 * hw]])
    d["masks"] = voronoi_masks(R, hw, g) if c.get("masks") == "voronoi10" else smooth_masks(R, hw, g)
    d["latents"] = torch.randn(1, 4, hw, hw, generator=g)                 # unscaled: prepare_latents multiplies by init_noise_sigma (xl.py:533-536)
    if c.get("unit_start"):
        sched = OracleEuler(); sched.set_timesteps(c["steps"])
        d["latents"] = d["latents"] / sched.init_noise_sigma                  # what the caller hands to `sample(latents=...)`
    tfd = {"word_pos": torch.tensor(c["word_pos"]), "font_size": torch.tensor(c["font_size"])}
    if c["guided"]:
        n = c["n_color"]
        tfd.update({"target_RGB": [torch.rand(1, 3, 1, 1, generator=g) for _ in range(n)], "guidance_start_step": 999,
                    "color_guidance_weight": c["color_weight"],
                    "color_obj_atten": [(torch.rand(1, 1, 8 * hw, 8 * hw, generator=g) ** 2).repeat(1, 4, 1, 1) for _ in range(n + 1)],
                    "color_obj_atten_all": torch.rand(1, 4, hw, hw, generator=g)})
    d["tfd"] = tfd
    return d


def unet_weights(name):
    c = CASES[name]
    return random_state_dict(SDXL_CONFIG if c["model"] == "sdxl" else SD15_CONFIG, seed=c["unet_seed"])


def vae_weights(name):
    c = CASES[name]
    return random_vae_state_dict(SDXL_VAE_CONFIG if c["model"] == "sdxl" else SD_VAE_CONFIG, seed=c["vae_seed"])


def to_uint8(img):
    """[1,3,H,W] decoder output in [-1,1] -> uint8 [H,W,3] (rd.py:234,267-271; VaeImageProcessor.postprocess for xl.py:943)."""
    return ((img / 2 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).round().to(torch.uint8)


def weights_fingerprint(sd):
    """Same digest as tests/oracle_cache.py (kept local: oracle/ does not import tests/)."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        t = sd[k].double()
        h.update(f"{k}|{tuple(t.shape)}|{float(t.sum()):.10g}|{float((t * t).sum()):.10g};".encode())
    return h.hexdigest()


def run_case(name):
    c = CASES[name]
    xl = c["model"] == "sdxl"
    t0 = time.perf_counter()
    usd, vsd = unet_weights(name), vae_weights(name)
    inp = case_inputs(name)
    unet = OracleUNet(SDXL_CONFIG if xl else SD15_CONFIG, usd)
    vcfg = SDXL_VAE_CONFIG if xl else SD_VAE_CONFIG
    vae = OracleVAEDecoder(vcfg, vsd)
    print(f"[{name}] weights + inputs drawn in {time.perf_counter() - t0:.0f} s", flush=True)
    m = inp["masks"]
    masks = [m[r:r + 1] for r in range(c["R"])]
    guidance = {"vae": vae, "scaling": vcfg["scaling_factor"]} if c["guided"] else None

    stop_after = c.get("stop_after")

    class Progress(list):                                                  # the loops append one latent per iteration
        def append(self, x):
            super().append(x)
            print(f"[{name}] iteration {len(self)} done at {time.perf_counter() - t0:.0f} s, latent std {x.std():.4f}", flush=True)
            if stop_after and len(self) == stop_after:
                raise StopLoop
    trace = Progress()
    try:
        if xl:
            sched = OracleEuler(); sched.set_timesteps(c["steps"])
            lat0 = inp["latents"] * sched.init_noise_sigma
            final = rich_loop_xl(unet, OracleEuler(), inp["emb"], inp["pooled"], inp["time_ids"], masks, lat0, c["steps"], c["gs"], inp["tfd"],
                                 c["isa"], c["ibg"], use_guidance=c["guided"], guidance=guidance, trace=trace)
        else:
            lat0 = inp["latents"]
            final = rich_loop_sd(unet, OraclePNDM(), inp["emb"], masks, lat0, c["steps"], c["gs"], inp["tfd"], c["isa"], c["ibg"],
                                 use_guidance=c["guided"], guidance=guidance, trace=trace)
    except StopLoop:
        final = trace[-1]
    assert torch.equal(final, trace[-1]) and len(trace) == c["checkpoints"][-1], (len(trace), c["checkpoints"])
    with torch.no_grad():
        image = to_uint8(vae.decode(final / vcfg["scaling_factor"]))
    out = {"case": dict(c), "unet_fingerprint": weights_fingerprint(usd), "vae_fingerprint": weights_fingerprint(vsd),
           "lat0": lat0, "checkpoints": {k: trace[k - 1].clone() for k in c["checkpoints"]}, "image_u8": image,
           "torch": str(torch.__version__), "oracle_seconds": time.perf_counter() - t0}
    os.makedirs(OUT, exist_ok=True)
    torch.save(out, os.path.join(OUT, name + ".pt"))
    print(f"[{name}] written; {time.perf_counter() - t0:.0f} s; final latent std {final.std():.4f}; image mean {image.float().mean():.1f}", flush=True)


def refingerprint(name):
    """Rewrites the weight fingerprints of an existing file (after a change of the digest's format) - the trajectories stay."""
    path = os.path.join(OUT, name + ".pt")
    d = torch.load(path, weights_only=False)
    d["unet_fingerprint"], d["vae_fingerprint"] = weights_fingerprint(unet_weights(name)), weights_fingerprint(vae_weights(name))
    d["torch"] = str(d["torch"])
    torch.save(d, path)
    print(f"[{name}] fingerprints rewritten")


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 1)))
    args = sys.argv[1:]
    if args and args[0] == "--refingerprint":
        for n in args[1:]:
            refingerprint(n)
    else:
        for n in args or ["config1", "config3", "config5"]:
            run_case(n)
