"""FULL-SCHEDULE parity at the architectures BASELINE.json names, in latents per checkpoint AND in final pixels.

tests/golden/fullschedule/<case>.pt holds trajectories of the fp32 CPU oracle (oracle/region_loop.py, pinned to the unmodified
reference loops by tests/test_oracle_vs_reference.py), generated in the build container by `python -m oracle.make_fullsize_golden`:

  * config1     RegionDiffusion.produce_latents      rd.py:86-174,227-236    SD-v1.5 @ 512x512, R = 2, 20 steps = 21 PLMS iterations
  * config2     the same loop + colour guidance        rd.py:151-168           SD-v1.5 @ 512x512, R = 4, 2 colour regions, 11 PLMS iterations
  * config3     RegionDiffusionXL.sample(rich)       xl.py:779-878,916-944   SDXL @ 1024x1024, R = 4, inject_selfattn 0.5, 10 Euler steps
  * config3_50  the same, the full 50-step schedule (the benched workload end to end)
  * config5     the same loop + colour guidance (xl.py:849-867) + background blend, CFG 7.5, 4 Euler steps @ 1024x1024
  * round 6: config2_50 (51 PLMS iterations), config5_50 (50-step schedule, 10-segment Voronoi masks, first 30 iterations across the
    blend at index 25), config3_unit (config3_50 from unit-variance latents: the update dominates the start noise)

The weights are not committed: `oracle.unet.random_state_dict(cfg, seed)` draws them with torch's CPU generator, bit-identically
here and in the build container (checked against the fingerprint in the file).  The HIP engine runs the same schedule through the
facade classes; after the iterations the file lists, the relative L2 error of the latents is asserted (the growth curve is printed and
written to gpurun_out/fullschedule_parity.json), and the final image is compared in uint8 pixels with the oracle's image (decoded by
the oracle VAE the way the reference decodes: /2 + 0.5, clamp, x255, round): PSNR, mean and max absolute difference, fraction of
pixels within 1 / 2 / 8 levels.  `decoder only` = the ORACLE's final latents through the engine's decoder: what the decoder alone
contributes.

Stated tolerances: TOL below (latents: relative L2 at every checkpoint; pixels: PSNR, mean and max absolute difference in uint8 levels).
"""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fullschedule_check as fc  # noqa: E402

# case: (max rel-L2 of the latents at ANY checkpoint, min PSNR [dB] of the final uint8 image, max mean-abs pixel difference, max-abs)
# Measured on MI355X (profiles/r5_fullschedule_parity.json): config1 4.9e-3 / 53.3 dB / 0.30 / 3; config3 4.1e-3 / 56.2 dB / 0.16 / 2;
# config5 8.2e-3 / 52.8 dB / 0.33 / 4; config3_50 (all 50 steps) 2.4e-3 / 58.7 dB / 0.09 / 1 - the error of one forward (7e-3 at CFG 1) does not compound over the schedule.
TOL = {
    "config1": (1.5e-2, 46.0, 1.0, 8),
    "config2": (1.5e-2, 46.0, 1.0, 8),
    "config3": (1.5e-2, 46.0, 1.0, 8),
    "config3_50": (1.5e-2, 46.0, 1.0, 8),
    "config5": (2.5e-2, 46.0, 1.0, 8),
    # round 6: the cases at BASELINE's own lengths / mask shape, and the unit-variance start (see UPDATE_TOL)
    "config2_50": (1.5e-2, 46.0, 1.0, 8),
    "config5_50": (2.5e-2, 46.0, 1.0, 8),
    # measured 5.4e-2 / 37.8 dB / 1.97 / 51 - and the same distance separates two summation orders of the engine itself (see the noise-floor test)
    "config3_unit": (8.0e-2, 34.0, 3.5, 96),
}
# The DISCRIMINATING bound: error relative to the accumulated update ||lat_k - lat_0|| of the oracle, at every recorded
# iteration.  With seeded random weights the UNet does not denoise: in the sigma-scaled cases the latents stay ~95 % start noise
# (||update|| / ||latents|| = 0.3 - 0.4), so the latent-relative figures above are 2.5 - 3x smaller than these; `config3_unit` starts
# from unit-variance latents, where the update IS the latent (ratio ~1) and the two figures coincide.  One forward differs from the
# fp32 oracle by 7e-3 (CFG 1); CFG multiplies that by the guidance scale relative to the difference of the two predictions.
# `config3_unit` measures 2.8e-2 after one step and 5.4e-2 from step 15 on: there the latents ARE the accumulated update, so an error in them
# perturbs the next UNet input by percents (not by 0.2 % as in the sigma-scaled cases) and the per-step errors add coherently instead of
# averaging out.  8e-2 = 1.5 x measured.
UPDATE_TOL = {"config1": 2.0e-2, "config2": 2.0e-2, "config3": 3.0e-2, "config3_50": 3.0e-2, "config5": 4.5e-2, "config2_50": 2.0e-2, "config5_50": 4.5e-2, "config3_unit": 8.0e-2}
RESULTS = {}


def _check(name, mdl, fp):
    if fc.load_golden(name) is None:
        pytest.skip(f"tests/golden/fullschedule/{name}.pt not generated (python -m oracle.make_fullsize_golden {name})")
    r = fc.compare(name, mdl, fp)
    RESULTS[name] = r
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "fullschedule_parity.json"), "w") as f:
        json.dump(RESULTS, f, indent=1)
    curve, pix, pix_dec = r["latent_rel_l2_by_iteration"], r["pixels_vs_oracle_image"], r["decoder_only"]
    ucurve = r["latent_update_rel_l2_by_iteration"]
    print(f"{name}: latent rel-L2 by loop iteration: " + ", ".join(f"{k}: {v:.3e}" for k, v in curve.items()))
    print(f"{name}: error relative to the accumulated UPDATE ||lat_k - lat_0||: " + ", ".join(f"{k}: {v:.3e}" for k, v in ucurve.items())
          + f"   (||update|| / ||latents|| at the end {r['update_over_latents_final']:.2f}; latent std {r['latent_std_start_final'][0]:.2f} -> {r['latent_std_start_final'][1]:.2f};"
          f" absolute rms error per element at the end {list(r['latent_abs_rms_by_iteration'].values())[-1]:.3e})")
    print(f"{name}: final image vs the oracle's image: PSNR {pix['psnr_db']:.2f} dB, mean |d| {pix['mean_abs']:.3f} / 255, max |d| {pix['max_abs']}, "
          f"within 1 / 2 / 8 levels {pix['within_1']:.4f} / {pix['within_2']:.4f} / {pix['within_8']:.4f}")
    print(f"{name}: decoder only (oracle latents through the engine's decoder): PSNR {pix_dec['psnr_db']:.2f} dB, mean |d| {pix_dec['mean_abs']:.4f}, max |d| {pix_dec['max_abs']}")
    t_lat, t_psnr, t_mean, t_max = TOL[name]
    assert max(curve.values()) < t_lat, curve
    assert max(ucurve.values()) < UPDATE_TOL[name], ucurve
    assert pix["psnr_db"] > t_psnr and pix["mean_abs"] < t_mean and pix["max_abs"] <= t_max, pix
    assert pix_dec["psnr_db"] > 40.0, pix_dec


@pytest.fixture(scope="module")
def sdxl_model():
    mdl, fp = fc.build_model("config3")
    yield mdl, fp
    fc.close_model(mdl)


@pytest.fixture(scope="module")
def sd_model():
    mdl, fp = fc.build_model("config1")
    yield mdl, fp
    fc.close_model(mdl)


def test_config3_ten_step_schedule_latents_and_pixels(sdxl_model):
    _check("config3", *sdxl_model)


def test_config3_full_fifty_step_schedule_latents_and_pixels(sdxl_model):
    _check("config3_50", *sdxl_model)


def test_config5_guided_schedule_latents_and_pixels(sdxl_model):
    _check("config5", *sdxl_model)


def test_config1_full_plms_schedule_latents_and_pixels(sd_model):
    _check("config1", *sd_model)


def test_config2_guided_plms_schedule_latents_and_pixels(sd_model):
    _check("config2", *sd_model)


def test_config2_full_fifty_step_guided_schedule_latents_and_pixels(sd_model):
    """BASELINE config 2 at its own length: 50 requested steps = 51 PLMS iterations, 4 regions, colour guidance on 2 of them."""
    _check("config2_50", *sd_model)


def test_config5_fifty_step_schedule_voronoi_masks_across_the_blend(sdxl_model):
    """BASELINE config 5 on its 50-step schedule with the 10-segment masks (10 Voronoi cells on 32 x 32 dealt to the 4 regions,
    resized / normalised as attention_utils.py:322-327), footnote + colour guidance, CFG 7.5, recorded over the first 30 iterations:
    across the background blend after loop index 25 (xl.py:870)."""
    _check("config5_50", *sdxl_model)


def test_config3_unit_variance_start_fifty_steps(sdxl_model):
    """config3_50 started from unit-variance latents: the accumulated update dominates the start noise, so the latent-relative error
    is the update-relative error - the regime a trained checkpoint ends in."""
    _check("config3_unit", *sdxl_model)


@pytest.mark.parametrize("name", ["config3", "config3_unit"])
def test_distance_from_the_oracle_is_the_rounding_order_noise_floor(sdxl_model, name):
    """What the full-schedule numbers measure (round 6, tools/trajectory_sensitivity.py, LABNOTES R6.3).  The engine is run over the schedule
    twice more with the SAME arithmetic in a different summation order (LayerNorm as launches instead of folded; the 32x32x16 GEMM family
    instead of the 16x16x32 one): bf16 operands and activations make any two orderings differ by a bf16 ulp per layer.  The engine's
    distance from the fp32 oracle must not exceed what separates those orderings from each other by more than a margin - a path with an
    arithmetic defect (as opposed to rounding) would stand out against this floor, in either regime."""
    import trajectory_sensitivity as ts
    if fc.load_golden(name) is None:
        pytest.skip(f"tests/golden/fullschedule/{name}.pt not generated")
    mdl, _ = sdxl_model
    gold = fc.load_golden(name)
    lat0 = gold["lat0"].float()
    ref = {k: v.float() for k, v in gold["checkpoints"].items()}
    base = ts.trajectory(name, mdl, 0)
    others = [ts.trajectory(name, mdl, f) for f in (1 << 22, 2)]
    worst = 0.0
    for k in sorted(ref):
        upd = (ref[k] - lat0).norm().item()
        d_oracle = (base[k] - ref[k]).norm().item() / upd
        floor = max((o[k] - base[k]).norm().item() for o in others) / upd
        print(f"{name} iteration {k}: engine vs fp32 oracle {d_oracle:.3e} | engine vs its own re-ordered arithmetic {floor:.3e}")
        assert floor > 0.0
        worst = max(worst, d_oracle / floor)
        assert d_oracle < 1.5 * floor + 2e-3, (k, d_oracle, floor)
    print(f"{name}: worst ratio (distance from the oracle) / (distance between two summation orders) = {worst:.2f}")


def test_one_pass_guidance_vae_leaves_the_guided_trajectory_where_the_precise_one_does(sdxl_model):
    """Round 6 option `RegionDiffusionXL.guidance_vae` (sample.py --guidance_precision bf16): the colour-guidance pass (xl.py:849-867) on a
    one-pass bf16 VaeDecoder while the final decode stays on the precise engine.  Over config 5's guided schedule the latents must stay within
    the SAME tolerances against the fp32 oracle, and within a hair of the precise-guidance run; the image (decoded precisely) keeps its bounds."""
    from oracle import make_fullsize_golden as mg
    from oracle.vae import SDXL_VAE_CONFIG
    from rich_text_to_image_amd.engine import VaeDecoder
    name = "config5"
    mdl, fp = sdxl_model
    base = RESULTS.get(name) or fc.compare(name, mdl, fp)
    c = mg.CASES[name]
    fast = VaeDecoder(SDXL_VAE_CONFIG, c["hw"], c["hw"], device=0, state_dict=mg.vae_weights(name), precise=False)
    mdl.guidance_vae = fast
    try:
        r = fc.compare(name, mdl, fp)
    finally:
        mdl.guidance_vae = None
        fast.close()
    t_lat, t_psnr, t_mean, t_max = TOL[name]
    u, ub = r["latent_update_rel_l2_by_iteration"], base["latent_update_rel_l2_by_iteration"]
    print(f"{name}: update-relative error with one-pass guidance {dict((k, round(v, 5)) for k, v in u.items())} | precise {dict((k, round(v, 5)) for k, v in ub.items())}")
    assert max(r["latent_rel_l2_by_iteration"].values()) < t_lat and max(u.values()) < UPDATE_TOL[name]
    assert all(abs(u[k] - ub[k]) < 0.1 * ub[k] + 1e-3 for k in u), (u, ub)
    pix = r["pixels_vs_oracle_image"]
    assert pix["psnr_db"] > t_psnr and pix["max_abs"] <= t_max, pix
