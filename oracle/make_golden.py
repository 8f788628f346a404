"""TEST INFRASTRUCTURE: generate tests/golden/*.pt by running the UNMODIFIED reference loops.

Runs only in the build container (needs /root/reference). It drives
  * RegionDiffusion.produce_latents        (/root/reference/models/region_diffusion.py:86-174)
  * RegionDiffusionXL.sample(run_rich_text) (/root/reference/models/region_diffusion_sdxl.py:555-953)
on tiny random-weight UNets (oracle.unet.TINY_*_CONFIG) with the restated schedulers
(oracle/schedulers.py; diffusers itself is not available), and stores inputs + outputs so that the
GPU-box tests can check (a) the oracle restatement and (b) the HIP engine against outputs of the
reference code itself.  Usage:  python -m oracle.make_golden
"""
import os
import sys
import types

import torch

from .refload import load_reference
from .schedulers import OracleEuler, OraclePNDM
from .unet import (TINY_SD_CONFIG, TINY_XL_CONFIG, OracleUNet, random_state_dict, reference_kwargs)
from . import region_loop

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def make_masks(R, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.softmax(torch.randn(R, 1, h // 4, w // 4, generator=g) * 4, dim=0)
    m = torch.nn.functional.interpolate(m, size=(h, w), mode="bilinear", align_corners=False)
    m = m / (m.sum(0, keepdim=True) + 1e-8)
    return [m[r:r + 1].repeat(1, 4, 1, 1).contiguous() for r in range(R)]


def case_inputs(cfg, xl, R, hw, seed):
    g = torch.Generator().manual_seed(seed)
    D = cfg["cross_attention_dim"]
    inp = {
        "latents": torch.randn(1, 4, hw, hw, generator=g),
        "embeds": torch.randn(R + 1, 77, D, generator=g),
        "masks": make_masks(R, hw, hw, seed + 1),
        "word_pos": torch.tensor([3, 5], dtype=torch.long),
        "font_size": torch.tensor([4.0, -2.0]),
    }
    if xl:
        inp["pooled"] = torch.randn(R + 1, 32, generator=g)
        inp["time_ids"] = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]])
    return inp


def run_reference_sd(mods, unet, inp, steps, gs, inject_selfattn, inject_background):
    RD = mods["region_diffusion"].RegionDiffusion
    m = RD.__new__(RD)
    torch.nn.Module.__init__(m)
    m.device = torch.device("cpu")
    m.unet = unet
    m.scheduler = OraclePNDM()
    m.masks = inp["masks"]
    unet.in_channels = 4
    tfd = {"word_pos": inp["word_pos"], "font_size": inp["font_size"]}
    with torch.no_grad():
        return m.produce_latents(inp["embeds"], height=inp["latents"].shape[2] * 8, width=inp["latents"].shape[3] * 8,
                                 num_inference_steps=steps, guidance_scale=gs, latents=inp["latents"].clone(),
                                 text_format_dict=tfd, inject_selfattn=inject_selfattn,
                                 inject_background=inject_background)


def run_reference_xl(mods, unet, inp, steps, gs, inject_selfattn, inject_background):
    XL = mods["region_diffusion_sdxl"].RegionDiffusionXL
    m = XL.__new__(XL)
    m.unet = unet
    m.scheduler = OracleEuler()
    m.device_type = "cpu"
    m.vae_scale_factor = 8
    m.default_sample_size = inp["latents"].shape[2]
    m.register_to_config(force_zeros_for_empty_prompt=True)
    m.tokenizer = m.tokenizer_2 = m.text_encoder = None
    m.text_encoder_2 = types.SimpleNamespace(config=types.SimpleNamespace(projection_dim=32))
    dummy_vae = types.SimpleNamespace(
        to=lambda **k: None,
        decoder=types.SimpleNamespace(mid_block=types.SimpleNamespace(attentions=[types.SimpleNamespace(processor=None)])))
    m.vae = dummy_vae
    m.masks = inp["masks"]
    m.check_inputs = lambda *a, **k: None
    sched = m.scheduler
    sched.set_timesteps(steps)
    lat = inp["latents"].clone()          # prepare_latents multiplies by init_noise_sigma (xl.py:536)
    tfd = {"word_pos": inp["word_pos"], "font_size": inp["font_size"]}
    hw = inp["latents"].shape[2] * 8
    out = m.sample(prompt=None, height=hw, width=hw, num_inference_steps=steps, guidance_scale=gs,
                   latents=lat, prompt_embeds=inp["embeds"][1:], negative_prompt_embeds=inp["embeds"][:1],
                   pooled_prompt_embeds=inp["pooled"][1:], negative_pooled_prompt_embeds=inp["pooled"][:1],
                   output_type="latent", run_rich_text=True, text_format_dict=tfd,
                   inject_selfattn=inject_selfattn, inject_background=inject_background,
                   original_size=(hw, hw), target_size=(hw, hw))
    return out.images


def main():
    os.makedirs(OUT, exist_ok=True)
    mods = load_reference()
    U = mods["unet_2d_condition"].UNet2DConditionModel
    AP = mods["attention_processor"]
    cases = [
        # name, cfg, xl, R, latent hw, steps, cfg scale, inject_selfattn, inject_background
        ("tiny_sd_plms", TINY_SD_CONFIG, False, 3, 64, 6, 7.5, 0.5, 0.5),
        ("tiny_sd_noinject", TINY_SD_CONFIG, False, 2, 64, 4, 8.5, 0.0, 0.0),
        ("tiny_xl_euler", TINY_XL_CONFIG, True, 3, 128, 6, 5.0, 0.5, 0.5),
        ("tiny_xl_bgonly", TINY_XL_CONFIG, True, 2, 128, 4, 7.5, 0.0, 0.5),
    ]
    for name, cfg, xl, R, hw, steps, gs, isa, ibg in cases:
        wseed = 11
        sd = random_state_dict(cfg, seed=wseed)
        ref = U(**reference_kwargs(cfg))
        ref.load_state_dict(sd)
        ref.eval()
        inp = case_inputs(cfg, xl, R, hw, seed=5)
        if xl:
            ref_out = run_reference_xl(mods, ref, inp, steps, gs, isa, ibg)
            sched = OracleEuler()
            sched.set_timesteps(steps)
            lat0 = inp["latents"] * sched.init_noise_sigma
            trace = []
            orc = region_loop.rich_loop_xl(OracleUNet(cfg, sd), OracleEuler(), inp["embeds"], inp["pooled"],
                                           inp["time_ids"], inp["masks"], lat0, steps, gs,
                                           {"word_pos": inp["word_pos"], "font_size": inp["font_size"]},
                                           isa, ibg, trace=trace)
        else:
            ref_out = run_reference_sd(mods, ref, inp, steps, gs, isa, ibg)
            trace = []
            orc = region_loop.rich_loop_sd(OracleUNet(cfg, sd), OraclePNDM(), inp["embeds"], inp["masks"],
                                           inp["latents"], steps, gs,
                                           {"word_pos": inp["word_pos"], "font_size": inp["font_size"]},
                                           isa, ibg, trace=trace)
        err = (ref_out - orc).abs().max().item()
        print(f"{name}: reference-vs-oracle max|diff| = {err:.3e}  (out std {ref_out.std().item():.3f})")
        assert err < 5e-4, name
        # single UNet forward golden (reference module output)
        with torch.no_grad():
            added = {"text_embeds": inp["pooled"][1:2], "time_ids": inp["time_ids"]} if xl else None
            unet_out = ref(inp["latents"], torch.tensor(481), encoder_hidden_states=inp["embeds"][1:2],
                           added_cond_kwargs=added)["sample"]
        torch.save({
            "name": name, "xl": xl, "R": R, "steps": steps, "guidance_scale": gs,
            "inject_selfattn": isa, "inject_background": ibg, "weight_seed": wseed,
            "weight_abs_sum": float(sum(v.abs().sum() for v in sd.values())),
            "inputs": {k: (v if not isinstance(v, list) else torch.cat(v)[:, :1].clone()) for k, v in inp.items()},
            "reference_final_latents": ref_out.clone(),
            "reference_unet_t481": unet_out.clone(),
            "oracle_trace_last": trace[-1].clone(),
        }, os.path.join(OUT, name + ".pt"))

    # operator-level golden: the reference Attention module (font-size + injection identities)
    torch.manual_seed(3)
    attn = AP.Attention(query_dim=64, cross_attention_dim=48, heads=2, dim_head=32)
    sattn = AP.Attention(query_dim=64, heads=2, dim_head=32)
    x = torch.randn(2, 256, 64)
    ctx = torch.randn(2, 77, 48)
    fs = {"word_pos": torch.tensor([2, 9, 30]), "font_size": torch.tensor([3.0, -1.5, 0.25])}
    with torch.no_grad():
        y_plain, _ = attn(x, encoder_hidden_states=ctx)
        y_fs, (_, p_fs) = attn(x, None, fs, encoder_hidden_states=ctx)
        y_self, (pavg, p_self) = sattn(x)
        x2 = torch.randn(2, 256, 64)
        y_inj, _ = sattn(x2, p_self)
    torch.save({
        "cross_sd": {k: v.clone() for k, v in attn.state_dict().items()},
        "self_sd": {k: v.clone() for k, v in sattn.state_dict().items()},
        "x": x, "ctx": ctx, "x2": x2, "word_pos": fs["word_pos"], "font_size": fs["font_size"],
        "y_plain": y_plain, "y_fs": y_fs, "p_fs_rowsum": p_fs.sum(-1), "y_self": y_self,
        "p_self_avg": pavg, "y_inj": y_inj,
    }, os.path.join(OUT, "attention_ops.pt"))
    print("wrote", sorted(os.listdir(OUT)))


def tokenmap_goldens():
    """Reference token-map hooks (rd.py:397-443 / xl.py:959-1016) + the reference get_token_maps
    (utils/attention_utils.py:233-341) on a plain pass of the tiny UNets."""
    import tempfile
    import matplotlib
    matplotlib.use("Agg")
    mods = load_reference()
    import importlib
    au = importlib.import_module("utils.attention_utils")
    au.plot_attention_maps = lambda *a, **k: None              # figures are out of scope
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self              # get_token_maps ends with .cuda() (attention_utils.py:337)
    U = mods["unet_2d_condition"].UNet2DConditionModel
    try:
        for name, cfg, xl, hw, steps in (("tokenmaps_sd", TINY_SD_CONFIG, False, 64, 13), ("tokenmaps_xl", TINY_XL_CONFIG, True, 128, 12)):
            sd = random_state_dict(cfg, seed=11)
            ref = U(**reference_kwargs(cfg)); ref.load_state_dict(sd); ref.eval()
            g = torch.Generator().manual_seed(9)
            lat = torch.randn(1, 4, hw, hw, generator=g)
            emb = torch.randn(2, 77, cfg["cross_attention_dim"], generator=g) * 2.0
            pooled = torch.randn(2, 32, generator=g) if xl else None
            tid = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * 2) if xl else None
            if xl:
                XL = mods["region_diffusion_sdxl"].RegionDiffusionXL
                m = XL.__new__(XL)
                # the XL hook reads the module-global list; point it at modules that exist in the tiny UNet
                xl_cross = ['down_blocks.2.attentions.1.transformer_blocks.1.attn2', 'mid_block.attentions.0.transformer_blocks.0.attn2',
                            'mid_block.attentions.0.transformer_blocks.1.attn2', 'up_blocks.0.attentions.0.transformer_blocks.1.attn2',
                            'up_blocks.1.attentions.0.transformer_blocks.0.attn2']
                mods["region_diffusion_sdxl"].CrossAttentionLayers_XL[:] = xl_cross
                sched = OracleEuler()
            else:
                RD = mods["region_diffusion"].RegionDiffusion
                m = RD.__new__(RD)
                torch.nn.Module.__init__(m)
                xl_cross = None
                sched = OraclePNDM()
            m.unet = ref
            m.register_tokenmap_hooks()
            sched.set_timesteps(steps)
            x = lat * sched.init_noise_sigma if xl else lat.clone()
            gs = 7.5
            for t in sched.timesteps:
                inp = torch.cat([x] * 2)
                if xl:
                    inp = sched.scale_model_input(inp, t)
                with torch.no_grad():
                    eps = ref(inp, t, encoder_hidden_states=emb,
                              added_cond_kwargs={"text_embeds": pooled, "time_ids": tid} if xl else None)["sample"]
                eu, et = eps.chunk(2)
                x = sched.step(eu + gs * (et - eu), t, x)["prev_sample"]
            selfm = {k: v for k, v in m.selfattn_maps.items() if v.shape[1] == 1024}
            crossm = dict(m.crossattn_maps)
            obj_tokens = [torch.tensor([2, 3]), torch.tensor([6])]
            with tempfile.TemporaryDirectory() as td:
                masks = au.get_token_maps(selfm, crossm, m.n_maps, td, hw, hw, obj_tokens, seed=3, segment_threshold=0.3, num_segments=5)
            print(name, "self maps", len(selfm), "cross maps", len(crossm), "n_maps", set(m.n_maps.values()), "masks", len(masks))
            torch.save({
                "xl": xl, "steps": steps, "guidance_scale": gs, "weight_seed": 11, "latents": lat, "embeds": emb, "pooled": pooled,
                "time_ids": tid[:1] if xl else None, "xl_cross_layers": xl_cross, "final_latents": x,
                "self_names": sorted(selfm), "cross_names": sorted(crossm),
                "self_maps_rows": {k: v[0, ::16].clone().half() for k, v in selfm.items()},      # every 16th query row (fixture size)
                "self_maps_rowsum": {k: v[0].sum(-1) for k, v in selfm.items()},
                "cross_maps": {k: v[0].clone() for k, v in crossm.items()},
                "n_maps_values": sorted(set(int(v) for v in m.n_maps.values())), "obj_tokens": obj_tokens,
                "masks": torch.cat(masks)[:, 0].clone().half(),
            }, os.path.join(OUT, name + ".pt"))
        # the reference get_token_maps on deterministic synthetic maps (pins the port; inputs are regenerated from the seed)
        from .synth import synthetic_attention_maps
        port = {}
        for seed, nseg, thr in ((0, 5, 0.3), (1, 7, 0.25)):
            selfm, crossm = synthetic_attention_maps(seed)
            with tempfile.TemporaryDirectory() as td:
                masks = au.get_token_maps(selfm, crossm, {}, td, 64, 64, [torch.tensor([2, 3]), torch.tensor([6])], seed=4,
                                          segment_threshold=thr, num_segments=nseg)
            port[(seed, nseg, thr)] = torch.cat(masks)[:, 0].clone()
        torch.save(port, os.path.join(OUT, "token_maps_port.pt"))
    finally:
        torch.Tensor.cuda = orig_cuda


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tokenmaps":
        tokenmap_goldens()
    else:
        main()
