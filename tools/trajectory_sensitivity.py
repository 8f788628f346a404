#!/usr/bin/env python3
"""How sensitive is a full-schedule trajectory to the ROUNDING ORDER of the arithmetic?  (round 6, LABNOTES R6.3)

  python tools/trajectory_sensitivity.py config3_unit config3_50

Runs the engine over the schedule of a committed oracle trajectory (tests/golden/fullschedule/<case>.pt) under several builds of the SAME
arithmetic that differ only in summation order / where a rounding happens (rt_op_gemm_debug switches: LayerNorm as launches instead of
folded, the 32x32x16 GEMM family instead of the 16x16x32 one, 3x3 convolutions on the patch kernel, no split-K) and prints, per recorded
iteration, the update-relative distance  ||a_k - b_k|| / ||oracle_k - lat_0||  (a) of every variant from the fp32 oracle and (b) between
the variants themselves.  If (b) is as large as (a), the number measures how the random-weight UNet amplifies ANY last-bit difference along
this schedule - a property of the test regime - and not an arithmetic defect of one path."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fullschedule_check as fc  # noqa: E402

VARIANTS = [("default", 0), ("LayerNorm launches (bit 22)", 1 << 22), ("gemm.hip 32x32x16 family (bit 1)", 2), ("3x3 convs on the patch kernel (bit 3)", 8),
            ("no split-K (bit 2)", 4)]


def trajectory(name, mdl, flags):
    from oracle import make_fullsize_golden as mg
    from rich_text_to_image_amd.engine import load_library
    lib = load_library()
    gold = fc.load_golden(name)
    c = gold["case"]
    inp = mg.case_inputs(name)
    m = inp["masks"]
    mdl.masks = [m[r:r + 1] for r in range(c["R"])]
    got = {}

    def cb(i, t, lat):
        if (i + 1) in gold["checkpoints"]:
            got[i + 1] = lat.cpu().float()
        if c.get("stop_after") and i + 1 == c["stop_after"]:
            raise mg.StopLoop
    lib.rt_op_gemm_debug(flags)
    try:
        emb, pooled = inp["emb"], inp["pooled"]
        mdl.sample(prompt=None, height=8 * c["hw"], width=8 * c["hw"], num_inference_steps=c["steps"], guidance_scale=c["gs"], latents=inp["latents"].clone(),
                   prompt_embeds=emb[1:], negative_prompt_embeds=emb[:1], pooled_prompt_embeds=pooled[1:], negative_pooled_prompt_embeds=pooled[:1],
                   output_type="latent", run_rich_text=True, text_format_dict=inp["tfd"], use_guidance=c["guided"], inject_selfattn=c["isa"],
                   inject_background=c["ibg"], callback=cb)
    except mg.StopLoop:
        pass
    finally:
        lib.rt_op_gemm_debug(0)
    return got


def main():
    names = sys.argv[1:] or ["config3_unit", "config3_50"]
    mdl, fp = fc.build_model(names[0])
    out = {}
    for name in names:
        gold = fc.load_golden(name)
        assert gold["case"]["model"] == "sdxl", "SDXL cases (the facade's sample() has the per-step callback)"
        lat0 = gold["lat0"].float()
        ref = {k: v.float() for k, v in gold["checkpoints"].items()}
        upd = {k: (ref[k] - lat0).norm().item() for k in ref}
        trajs = {label: trajectory(name, mdl, flags) for label, flags in VARIANTS}
        ks = sorted(ref)
        rec = {"vs_oracle": {}, "between_variants": {}}
        print(f"== {name}: distance / ||oracle_k - lat_0|| at iterations {ks}")
        for label, t in trajs.items():
            rec["vs_oracle"][label] = {k: (t[k] - ref[k]).norm().item() / upd[k] for k in ks}
            print(f"  {label:42s} vs fp32 oracle : " + " ".join(f"{rec['vs_oracle'][label][k]:.2e}" for k in ks))
        base = trajs["default"]
        for label, t in trajs.items():
            if label == "default":
                continue
            rec["between_variants"][label] = {k: (t[k] - base[k]).norm().item() / upd[k] for k in ks}
            print(f"  {label:42s} vs default     : " + " ".join(f"{rec['between_variants'][label][k]:.2e}" for k in ks))
        out[name] = rec
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "trajectory_sensitivity.json"), "w") as f:
        json.dump(out, f, indent=1)
    fc.close_model(mdl)


if __name__ == "__main__":
    main()
