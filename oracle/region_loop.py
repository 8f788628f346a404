"""ORACLE (test infrastructure): the rich-text denoising loops restated on top of OracleUNet.

  * rich_loop_sd  follows RegionDiffusion.produce_latents      models/region_diffusion.py:86-174
  * rich_loop_xl  follows RegionDiffusionXL.sample (rich branch) models/region_diffusion_sdxl.py:772-878
                  (+ prepare_latents scaling :533-536 is the caller's job: pass already-scaled latents)

Colour guidance (rd.py:151-168 / xl.py:849-867) needs AutoencoderKL (third-party, not on disk);
it is restated in oracle/vae.py and enabled by passing `guidance=`.
Pinned against the unmodified reference loops in tests/test_oracle_vs_reference.py.
"""
import torch

from .unet import INJECT_RESNET


def _fontsize(tfd):
    if tfd and tfd.get("word_pos") is not None and tfd.get("font_size") is not None:
        return {"word_pos": tfd["word_pos"], "font_size": tfd["font_size"]}
    return None


def predict_x0(sched, x_t, eps_t, t):
    a = sched.alphas_cumprod[int(t)]
    return (x_t - eps_t * torch.sqrt(1 - a)) / torch.sqrt(a)


def _guidance_update(latents, noise_pred, t, sched, guidance, tfd):
    """rd.py:151-168 / xl.py:849-867. guidance = {"vae": OracleVAEDecoder, "scaling": s}."""
    lat = latents.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        x0 = predict_x0(sched, lat, noise_pred, t)
        imgs = guidance["vae"].decode(x0 / guidance["scaling"])
        imgs = (imgs / 2 + 0.5).clamp(0, 1)
        loss_total = 0.0
        for attn_map, rgb in zip(tfd["color_obj_atten"], tfd["target_RGB"]):
            avg = (imgs * attn_map[:, 0]).sum(2).sum(2) / attn_map[:, 0].sum()
            loss_total = loss_total + torch.nn.functional.mse_loss(avg, rgb[:, :, 0, 0]) * 100
        loss_total.backward()
    return (lat - lat.grad * tfd["color_guidance_weight"] * tfd["color_obj_atten_all"]).detach().clone()


def rich_step_forwards(unet, lat_in, lat_ref_in, t, embeds, added_fn, masks, tfd, use_ref, feat_inject_step):
    """Steps 1-5 of SURVEY 3.2: all UNet forwards of one rich step + mask combine.
    Returns (eps_uncond, eps_text, eps_u_ref, eps_t_ref)."""
    with torch.no_grad():
        eps_u = unet.forward(lat_in, t, embeds[:1], added_fn(0))
        eps_b = unet.forward(lat_in, t, embeds[-1:], added_fn(-1), ctl={"fontsize": _fontsize(tfd)})
        eps_ur = eps_tr = None
        cap = {}
        if use_ref:
            eps_ur = unet.forward(lat_ref_in, t, embeds[:1], added_fn(0))
            eps_tr = unet.forward(lat_ref_in, t, embeds[-1:], added_fn(-1),
                                  ctl={"capture": cap if feat_inject_step else None})
        eps_uncond = eps_u * masks[-1]
        eps_text = eps_b * masks[-1]
        for r, mask in enumerate(masks[:-1]):
            inj = None
            if feat_inject_step:
                # register_replacement_hooks: attn1 probs + the one resnet feature (xl.py:1018-1062)
                inj = {k: v for k, v in cap.items() if k.endswith("attn1") or k == INJECT_RESNET}
            eps_r = unet.forward(lat_in, t, embeds[r + 1:r + 2], added_fn(r + 1), ctl={"inject": inj})
            eps_uncond = eps_uncond + eps_u * mask
            eps_text = eps_text + eps_r * mask
    return eps_uncond, eps_text, eps_ur, eps_tr


def rich_loop_sd(unet, sched, text_embeddings, masks, latents, num_inference_steps, guidance_scale=7.5,
                 text_format_dict=None, inject_selfattn=0, inject_background=0, use_guidance=False,
                 guidance=None, trace=None):
    tfd = text_format_dict or {}
    use_ref = inject_selfattn > 0 or inject_background > 0
    lat = latents.clone()
    lat_ref = latents.clone() if use_ref else None
    sched.set_timesteps(num_inference_steps)
    assert text_embeddings.shape[0] - 1 == len(masks)                       # rd.py:97
    n = len(sched.timesteps)
    for i, t in enumerate(sched.timesteps):
        feat = bool(t > (1 - inject_selfattn) * 1000)                       # rd.py:104
        bg = (i == int(inject_background * n)) and inject_background > 0    # rd.py:105
        eu, et, eur, etr = rich_step_forwards(unet, lat, lat_ref, t, text_embeddings, lambda k: None,
                                              masks, tfd, use_ref, feat)
        eps = eu + guidance_scale * (et - eu)
        if use_ref:
            eps_ref = eur + guidance_scale * (etr - eur)
            out = sched.step(torch.cat([eps, eps_ref]), t, torch.cat([lat, lat_ref]))["prev_sample"]
            lat, lat_ref = torch.chunk(out, 2, dim=0)
        else:
            lat = sched.step(eps, t, lat)["prev_sample"]
        if use_guidance and t < tfd["guidance_start_step"]:
            lat = _guidance_update(lat, eps, t, sched, guidance, tfd)
        if bg:
            lat = lat_ref * masks[-1] + lat * (1 - masks[-1])
        if trace is not None:
            trace.append(lat.clone())
    return lat


def rich_loop_xl(unet, sched, prompt_embeds, add_text_embeds, add_time_ids, masks, latents,
                 num_inference_steps, guidance_scale=5.0, text_format_dict=None, inject_selfattn=0,
                 inject_background=0, use_guidance=False, guidance=None, trace=None):
    """prompt_embeds [R+1,77,D] = cat(negative, region prompts..., base) (xl.py:760-762);
    `latents` already multiplied by init_noise_sigma (prepare_latents, xl.py:536)."""
    tfd = text_format_dict or {}
    use_ref = inject_selfattn > 0 or inject_background > 0
    sched.set_timesteps(num_inference_steps)
    lat = latents.clone()
    lat_ref = latents.clone() if use_ref else None
    n = len(sched.timesteps)

    def added_fn(k):
        k = k if k >= 0 else add_text_embeds.shape[0] + k
        return {"text_embeds": add_text_embeds[k:k + 1], "time_ids": add_time_ids[:1]}

    for i, t in enumerate(sched.timesteps):
        feat = bool(t > (1 - inject_selfattn) * 1000)                       # xl.py:782
        bg_step = i < inject_background * n                                  # xl.py:783
        lat_in = sched.scale_model_input(lat, t)
        lat_ref_in = sched.scale_model_input(lat_ref, t) if use_ref else None
        eu, et, eur, etr = rich_step_forwards(unet, lat_in, lat_ref_in, t, prompt_embeds, added_fn,
                                              masks, tfd, use_ref, feat)
        eps = eu + guidance_scale * (et - eu)
        if inject_selfattn > 0 or bg_step > 0:                               # xl.py:832
            eps_ref = eur + guidance_scale * (etr - eur)
            out = sched.step(torch.cat([eps, eps_ref]), t, torch.cat([lat, lat_ref]))["prev_sample"]
            lat, lat_ref = torch.chunk(out, 2, dim=0)
        else:
            lat = sched.step(eps, t, lat)["prev_sample"]
        if use_guidance and t < tfd["guidance_start_step"]:
            lat = _guidance_update(lat, eps, t, sched, guidance, tfd)
        if i == int(inject_background * n) and inject_background > 0:        # xl.py:870
            lat = lat_ref * masks[-1] + lat * (1 - masks[-1])
        if trace is not None:
            trace.append(lat.clone())
    return lat


def plain_loop(unet, sched, embeds, latents, num_inference_steps, guidance_scale, added=None, xl=False,
               store=None):
    """Plain-text pass (rd.py:180-225 / xl.py:879-914): batch-2 CFG forwards; `store` receives the
    per-head probabilities of every attention module (token-map hooks)."""
    sched.set_timesteps(num_inference_steps)
    lat = latents.clone()
    for t in sched.timesteps:
        x = torch.cat([lat] * 2)
        if xl:
            x = sched.scale_model_input(x, t)
        with torch.no_grad():
            eps = unet.forward(x, t, embeds, added, store=store)
        eu, et = eps.chunk(2)
        eps = eu + guidance_scale * (et - eu)
        lat = sched.step(eps, t, lat)["prev_sample"]
    return lat
