"""End-to-end flow of the reference's `sample.py:17-114` on the MI355X engine:
JSON -> spans -> plain pass with on-device attention capture -> token maps (twice) -> rich-text pass.

`generate(model, param, ...)` is the body of the reference `main()` with the model passed in (checkpoints, the CLIP
tokenizer and text encoders are the caller's: none are available offline); file output is optional.
"""
import argparse
import json
import os
import time

import torch
import torch.nn.functional as F

from .attention_utils import get_token_maps
from .richtext_utils import (get_attention_control_input, get_gradient_guidance_input, get_region_diffusion_input, parse_json,
                             seed_everything)


def _resize_bicubic_aa(mask, height, width):
    # torchvision.transforms.functional.resize(tensor, BICUBIC, antialias=True) == this interpolate call (sample.py:83-86)
    return F.interpolate(mask, size=(height, width), mode='bicubic', antialias=True, align_corners=False)


def generate(model, param, model_type='SD', run_dir=None, color_guidance_weight=0.5, inject_selfattn=0., segment_threshold=0.3,
             num_segments=9, inject_background=0., latents=None):
    """Returns (plain_image, rich_image, timings).  `param` has the reference's keys: text_input, height, width,
    guidance_weight, steps, noise_index, negative_prompt (sample.py:135-143)."""
    if run_dir:
        os.makedirs(run_dir, exist_ok=True)
    spans = parse_json(param['text_input'], device=model.device)
    base_text_prompt, style_p, note_p, note_tok, color_p, color_names, color_rgbs, sizes, use_grad_guidance = spans
    region_text_prompts, region_target_token_ids, base_tokens = get_region_diffusion_input(
        model, base_text_prompt, style_p, note_p, note_tok, color_p, color_names)
    text_format_dict = get_attention_control_input(model, base_tokens, sizes, device=model.device)
    text_format_dict, color_target_token_ids = get_gradient_guidance_input(
        model, base_tokens, color_p, color_rgbs, text_format_dict, color_guidance_weight=color_guidance_weight)
    height, width, seed, negative_text = param['height'], param['width'], param['noise_index'], param['negative_prompt']
    timings = {}

    seed_everything(seed)
    t0 = time.time()
    if model.attention_maps is None:
        model.register_tokenmap_hooks()
    else:
        model.reset_attention_maps()
    if model_type == 'SD':
        plain_img = model.produce_attn_maps([base_text_prompt], [negative_text], height=height, width=width,
                                            num_inference_steps=param['steps'], guidance_scale=param['guidance_weight'], latents=latents)
    else:
        plain_img = model.sample([base_text_prompt], negative_prompt=[negative_text], height=height, width=width,
                                 num_inference_steps=param['steps'], guidance_scale=param['guidance_weight'], run_rich_text=False,
                                 latents=latents)
    timings['plain'] = time.time() - t0

    t0 = time.time()
    seed_everything(seed)
    color_obj_masks = get_token_maps(model.selfattn_maps, model.crossattn_maps, model.n_maps, run_dir, height // 8, width // 8,
                                     color_target_token_ids[:-1], seed, base_tokens, segment_threshold=segment_threshold,
                                     num_segments=num_segments)
    color_obj_atten_all = torch.zeros_like(color_obj_masks[-1])
    for m in color_obj_masks[:-1]:
        color_obj_atten_all += m
    text_format_dict['color_obj_atten'] = [_resize_bicubic_aa(m, height, width) for m in color_obj_masks]
    text_format_dict['color_obj_atten_all'] = color_obj_atten_all
    seed_everything(seed)
    model.masks = get_token_maps(model.selfattn_maps, model.crossattn_maps, model.n_maps, run_dir, height // 8, width // 8,
                                 region_target_token_ids[:-1], seed, base_tokens, segment_threshold=segment_threshold,
                                 num_segments=num_segments)
    model.remove_tokenmap_hooks()
    timings['token_maps'] = time.time() - t0

    t0 = time.time()
    seed_everything(seed)
    if model_type == 'SD':
        rich_img = model.prompt_to_img(region_text_prompts, [negative_text], height=height, width=width,
                                       num_inference_steps=param['steps'], guidance_scale=param['guidance_weight'],
                                       use_guidance=use_grad_guidance, inject_selfattn=inject_selfattn,
                                       text_format_dict=text_format_dict, inject_background=inject_background, latents=latents)
    else:
        rich_img = model.sample(region_text_prompts, negative_prompt=[negative_text], height=height, width=width,
                                num_inference_steps=param['steps'], guidance_scale=param['guidance_weight'],
                                use_guidance=use_grad_guidance, inject_selfattn=inject_selfattn, text_format_dict=text_format_dict,
                                inject_background=inject_background, run_rich_text=True, latents=latents)
    timings['rich'] = time.time() - t0
    return plain_img, rich_img, timings


def main(argv=None):
    """Flags of sample.py:118-133 (+ --load_path).  The model is constructed exactly as sample.py:24-32 does."""
    p = argparse.ArgumentParser()
    p.add_argument('--run_dir', type=str, default='results/')
    p.add_argument('--height', type=int, default=None)
    p.add_argument('--width', type=int, default=None)
    p.add_argument('--seed', type=int, default=6)
    p.add_argument('--sample_steps', type=int, default=41)
    p.add_argument('--rich_text_json', type=str, required=True)
    p.add_argument('--negative_prompt', type=str, default='')
    p.add_argument('--model', type=str, default='SD', choices=['SD', 'SDXL', 'AnimeXL'])
    p.add_argument('--guidance_weight', type=float, default=8.5)
    p.add_argument('--color_guidance_weight', type=float, default=0.5)
    p.add_argument('--inject_selfattn', type=float, default=0.)
    p.add_argument('--segment_threshold', type=float, default=0.3)
    p.add_argument('--num_segments', type=int, default=9)
    p.add_argument('--inject_background', type=float, default=0.)
    p.add_argument('--load_path', type=str, default=None,
                   help='diffusers-layout checkpoint directory; default: the hub ids of sample.py:26-30 resolved locally '
                        '(checkpoint.resolve_checkpoint: $RTDIFF_SD_PATH / $RTDIFF_SDXL_PATH / the Hugging Face hub cache)')
    a = p.parse_args(argv)
    from .region_diffusion import RegionDiffusion
    from .region_diffusion_sdxl import RegionDiffusionXL
    res = 512 if a.model == 'SD' else 1024
    # the VAE plan (decode + colour guidance workspace) is sized for the requested image, like the UNet engine
    hw = ((a.height or res) // 8, (a.width or res) // 8)
    # sample.py:24-32
    if a.model == 'SD':
        device = torch.device('cuda')
        model = RegionDiffusion(device, load_path=a.load_path, latent_hw=hw)
    elif a.model == 'SDXL':
        model = RegionDiffusionXL(load_path=a.load_path or "stabilityai/stable-diffusion-xl-base-1.0", latent_hw=hw)
    else:
        model = RegionDiffusionXL(load_path=a.load_path or "Linaqruf/animagine-xl", latent_hw=hw)
    param = {'text_input': json.loads(a.rich_text_json), 'height': a.height or res, 'width': a.width or res,
             'guidance_weight': a.guidance_weight, 'steps': a.sample_steps, 'noise_index': a.seed, 'negative_prompt': a.negative_prompt}
    plain, rich, t = generate(model, param, 'SD' if a.model == 'SD' else 'SDXL', a.run_dir, a.color_guidance_weight, a.inject_selfattn,
                              a.segment_threshold, a.num_segments, a.inject_background)
    print('time lapses: plain %.3f s, token maps %.3f s, rich %.3f s' % (t['plain'], t['token_maps'], t['rich']))
    # seed%d_plain.jpg / seed%d_rich.jpg in run_dir, as sample.py:62-76,97-112 writes them (imageio there, PIL here)
    from PIL import Image
    for name, img in (('plain', plain), ('rich', rich)):
        arr = img.images[0] if hasattr(img, 'images') else img[0]
        (arr if isinstance(arr, Image.Image) else Image.fromarray(arr)).save(os.path.join(a.run_dir, 'seed%d_%s.jpg' % (a.seed, name)))
    return plain, rich


if __name__ == '__main__':
    main()
