"""End-to-end flow of the reference's `sample.py:17-114` on the MI355X engine:
JSON -> spans -> plain pass with on-device attention capture -> token maps (twice) -> rich-text pass.

`generate(model, param, ...)` is the body of the reference `main()` with the model passed in (checkpoints, the CLIP
tokenizer and text encoders are the caller's: none are available offline); file output is optional.
"""
import argparse
import json
import os
import sys
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
    split = bool(getattr(model, 'split_image', False))
    capture_rank, my_rank = 0, 0
    if split:
        # --split_image: the plain pass ran one stream per rank (launcher.split_plain_step), so only the rank of the TEXT stream holds the
        # recorded maps: it derives the masks and hands them to the others (a few hundred KB) - every rank then holds identical masks by
        # construction (the rich pass still digests them, region_diffusion*.py: assert_ranks_agree)
        import torch.distributed as dist
        from .launcher import broadcast_objects, plain_capture_rank
        capture_rank, my_rank = plain_capture_rank(), (dist.get_rank() if dist.is_initialized() else 0)
    if my_rank == capture_rank:
        seg_cache = {}       # both calls cluster the same recorded maps with the same seed: the second reuses the first's segmentation
        color_obj_masks = get_token_maps(model.selfattn_maps, model.crossattn_maps, model.n_maps, run_dir, height // 8, width // 8,
                                         color_target_token_ids[:-1], seed, base_tokens, segment_threshold=segment_threshold,
                                         num_segments=num_segments, cache=seg_cache)
        seed_everything(seed)
        region_masks = get_token_maps(model.selfattn_maps, model.crossattn_maps, model.n_maps, run_dir, height // 8, width // 8,
                                      region_target_token_ids[:-1], seed, base_tokens, segment_threshold=segment_threshold,
                                      num_segments=num_segments, cache=seg_cache)
    else:
        color_obj_masks = region_masks = None
    if split:
        dev = None if color_obj_masks is None else color_obj_masks[-1].device
        packed = None if color_obj_masks is None else ([m.cpu() for m in color_obj_masks], [m.cpu() for m in region_masks])
        color_obj_masks, region_masks = broadcast_objects(packed, capture_rank)
        dev = dev if dev is not None else getattr(model, 'device', 'cpu')
        color_obj_masks, region_masks = [m.to(dev) for m in color_obj_masks], [m.to(dev) for m in region_masks]
    color_obj_atten_all = torch.zeros_like(color_obj_masks[-1])
    for m in color_obj_masks[:-1]:
        color_obj_atten_all += m
    text_format_dict['color_obj_atten'] = [_resize_bicubic_aa(m, height, width) for m in color_obj_masks]
    text_format_dict['color_obj_atten_all'] = color_obj_atten_all
    seed_everything(seed)
    model.masks = region_masks
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


def _load_json_arg(v):
    """`--rich_text_json` value: the JSON text itself (as the reference passes it, sample.py:123) or a path to a file holding it."""
    if os.path.isfile(v):
        with open(v) as f:
            return json.load(f)
    return json.loads(v)


def build_requests(a):
    """The (rich-text JSON, seed) list of one invocation.  One JSON + one --seed = the reference's single image.  Several JSONs and / or
    several --seeds = independent requests (BASELINE configs 4 / 5: "batch of 8 independent rich-text JSON/seeds"): JSONs and seeds
    pair up one to one when both lists have the same length, otherwise every JSON is sampled with every seed.  `--requests FILE`:
    one JSON object per line, {"rich_text_json": {...} | "<path>", "seed": 3, "negative_prompt": "..."} (missing keys: the flags)."""
    reqs = []
    if a.requests:
        with open(a.requests) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                r = json.loads(line)
                js = r["rich_text_json"]
                reqs.append(dict(text_input=_load_json_arg(js) if isinstance(js, str) else js, seed=int(r.get("seed", a.seed)),
                                 negative_prompt=r.get("negative_prompt", a.negative_prompt)))
    jsons = [_load_json_arg(v) for v in (a.rich_text_json or [])]
    seeds = list(a.seeds) if a.seeds else [a.seed]
    if jsons:
        pairs = list(zip(jsons, seeds)) if len(jsons) == len(seeds) else [(j, sd) for j in jsons for sd in seeds]
        reqs += [dict(text_input=j, seed=int(sd), negative_prompt=a.negative_prompt) for j, sd in pairs]
    if not reqs:
        raise SystemExit("sample: no request (pass --rich_text_json JSON [JSON ...] and / or --requests FILE)")
    for i, r in enumerate(reqs):
        r["index"] = i
    return reqs


def build_model(a, rank=0, local_rank=0, world=1):
    """sample.py:24-32.  world > 1: rank 0 reads the checkpoint, the other ranks build the same objects from the config / tokenizer
    files alone and receive the packed UNet arena, the VAE arena and the text-encoder weights in launcher.broadcast_pipeline's three
    collectives (RCCL over xGMI; SURVEY 8e: one broadcast at start-up, no per-step collectives)."""
    from .region_diffusion import RegionDiffusion
    from .region_diffusion_sdxl import RegionDiffusionXL
    res = 512 if a.model == 'SD' else 1024
    # the VAE plan (decode + colour guidance workspace) is sized for the requested image, like the UNet engine
    hw = ((a.height or res) // 8, (a.width or res) // 8)
    if world == 1:
        if a.model == 'SD':
            return RegionDiffusion(torch.device('cuda'), load_path=a.load_path, latent_hw=hw), 0.0
        xl_path = a.load_path or ("stabilityai/stable-diffusion-xl-base-1.0" if a.model == 'SDXL' else "Linaqruf/animagine-xl")
        model = RegionDiffusionXL(load_path=xl_path, latent_hw=hw)
        if getattr(a, 'guidance_precision', 'fp32') == 'bf16':
            from .checkpoint import load_guidance_vae, resolve_checkpoint
            model.guidance_vae = load_guidance_vae(resolve_checkpoint(xl_path, 'SDXL'), 'SDXL', 0, hw)
        return model, 0.0
    from . import launcher
    from .checkpoint import load_components, resolve_checkpoint
    kind = 'SD' if a.model == 'SD' else 'SDXL'
    default = None if a.model != 'AnimeXL' else "Linaqruf/animagine-xl"
    comp = load_components(resolve_checkpoint(a.load_path or default, kind), kind, local_rank, hw, weights=(rank == 0))
    if kind == 'SD':
        model = RegionDiffusion(local_rank, **comp)
        encoders = [comp["text_encoder"].enc]
    else:
        model = RegionDiffusionXL(None, local_rank, **comp)
        encoders = list(comp["text_encoders"].encoders)
    eng = model.unet.engine(hw[0], hw[1])                            # the engine (and its arena) must exist on every rank before the broadcast
    seconds = launcher.broadcast_pipeline(eng, model.vae, encoders, src=0)
    return model, seconds


def dist_backend_is_gloo():
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() == "gloo"


def main(argv=None):
    """Flags of sample.py:118-133 (+ --load_path), and the seed-parallel form the reference does not have (BASELINE
    configs 4 / 5): `--gpus N` with several requests (`--rich_text_json A B ...`, `--seeds ...`, `--requests FILE`) re-executes itself
    as N ranks under torch.distributed.run, rank 0 loads the checkpoint, ONE pipeline broadcast, requests dealt round-robin, every
    rank writes its own images.  With one GPU and one request this is the reference's main()."""
    import sys
    p = argparse.ArgumentParser()
    p.add_argument('--run_dir', type=str, default='results/')
    p.add_argument('--height', type=int, default=None)
    p.add_argument('--width', type=int, default=None)
    p.add_argument('--seed', type=int, default=6)
    p.add_argument('--sample_steps', type=int, default=41)
    p.add_argument('--rich_text_json', type=str, nargs='+', default=None, help='JSON text or a file holding it; several = independent requests')
    p.add_argument('--negative_prompt', type=str, default='')
    p.add_argument('--model', type=str, default='SD', choices=['SD', 'SDXL', 'AnimeXL'])
    p.add_argument('--guidance_weight', type=float, default=8.5)
    p.add_argument('--color_guidance_weight', type=float, default=0.5)
    p.add_argument('--inject_selfattn', type=float, default=0.)
    p.add_argument('--segment_threshold', type=float, default=0.3)
    p.add_argument('--num_segments', type=int, default=9)
    p.add_argument('--inject_background', type=float, default=0.)
    p.add_argument('--guidance_precision', type=str, default='fp32', choices=['fp32', 'bf16'],
                   help='SDXL colour guidance: fp32 = the fp32-class VAE the reference guides with (xl.py:856; default), bf16 = the guidance pass '
                        'alone on a one-pass bf16 VAE engine (same trajectory within the bf16 noise of the UNet, 37 instead of 80 ms per step); '
                        'the final decode stays fp32-class either way')
    p.add_argument('--load_path', type=str, default=None,
                   help='diffusers-layout checkpoint directory; default: the hub ids of sample.py:26-30 resolved locally '
                        '(checkpoint.resolve_checkpoint: $RTDIFF_SD_PATH / $RTDIFF_SDXL_PATH / the Hugging Face hub cache)')
    p.add_argument('--seeds', type=int, nargs='*', default=None, help='seeds of the independent requests (default: --seed)')
    p.add_argument('--requests', type=str, default=None, help='JSON-lines request file (see build_requests)')
    p.add_argument('--gpus', type=int, default=1, help='ranks = GPUs of this node; requests are dealt round-robin (one engine per GPU)')
    p.add_argument('--split_image', action='store_true',
                   help='with --gpus N: the ranks work on the SAME request - every rich-text step is split by streams (text_ref and the '
                        'region forwards that inject from it on one GPU, the independent forwards on the others), one exchange of the noise '
                        'predictions per step (launcher.split_region_step): lower latency per image instead of more images; bit-identical '
                        'with the one-GPU result.  Rank 0 writes the images.')
    p.add_argument('--dry_launch', action='store_true',
                   help='exercise the launch / sharding / broadcast control path on the gloo backend with stand-in arenas and print '
                        'one JSON line per rank - no GPU, no checkpoint (tests/test_distributed_cpu.py)')
    argv = list(sys.argv[1:] if argv is None else argv)
    a = p.parse_args(argv)
    from . import launcher
    err = launcher.self_launch(None, argv, a.gpus, require_gpus=not a.dry_launch, module=__spec__.name if __spec__ else "rich_text_to_image_amd.sample")
    if err is not None:
        raise SystemExit(f"sample: {err}")
    # RTDIFF_DIST_BACKEND / RTDIFF_FORCE_DEVICE: tests that run two ranks on ONE GPU over gloo (RCCL cannot put two ranks on a device)
    rank, local_rank, world = launcher.init_distributed("gloo" if a.dry_launch else os.environ.get("RTDIFF_DIST_BACKEND"))
    if world > 1:                     # prove the collective path before the pipeline's weights move (launcher.collective_self_check)
        chk = launcher.collective_self_check(nbytes=(1 << 20) if a.dry_launch else (64 << 20),
                                             device=torch.device("cpu") if (a.dry_launch or dist_backend_is_gloo()) else None)
        if rank == 0:
            print(f"sample: collective self-check {chk}", file=sys.stderr, flush=True)
    if "RTDIFF_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["RTDIFF_FORCE_DEVICE"])
    if world != max(1, a.gpus) and "WORLD_SIZE" in os.environ:
        raise SystemExit(f"sample: --gpus {a.gpus} but the launch environment says WORLD_SIZE={world}")
    reqs = build_requests(a)
    mine = reqs if a.split_image else launcher.shard_round_robin(reqs, rank, world)      # --split_image: every rank runs every request
    if a.dry_launch:
        return _dry_launch(a, rank, world, reqs, mine)
    torch.cuda.set_device(local_rank)
    model, bcast_s = build_model(a, rank, local_rank, world)
    res = 512 if a.model == 'SD' else 1024
    model.split_image = bool(a.split_image and world > 1)
    os.makedirs(a.run_dir, exist_ok=True)
    from PIL import Image
    out = []
    for r in mine:
        param = {'text_input': r['text_input'], 'height': a.height or res, 'width': a.width or res, 'guidance_weight': a.guidance_weight,
                 'steps': a.sample_steps, 'noise_index': r['seed'], 'negative_prompt': r['negative_prompt']}
        plain, rich, t = generate(model, param, 'SD' if a.model == 'SD' else 'SDXL', a.run_dir, a.color_guidance_weight, a.inject_selfattn,
                                  a.segment_threshold, a.num_segments, a.inject_background)
        print('[rank %d] request %d seed %d: time lapses: plain %.3f s, token maps %.3f s, rich %.3f s'
              % (rank, r['index'], r['seed'], t['plain'], t['token_maps'], t['rich']), flush=True)
        # seed%d_plain.jpg / seed%d_rich.jpg in run_dir, as sample.py:62-76,97-112 writes them (imageio there, PIL here); with several
        # requests the request index keeps the names apart
        tag = 'seed%d' % r['seed'] if len(reqs) == 1 else 'req%d_seed%d' % (r['index'], r['seed'])
        for name, img in (('plain', plain), ('rich', rich)):
            if a.split_image and rank != 0:           # every rank holds the same image: rank 0 writes it
                continue
            arr = img.images[0] if hasattr(img, 'images') else img[0]
            (arr if isinstance(arr, Image.Image) else Image.fromarray(arr)).save(os.path.join(a.run_dir, '%s_%s.jpg' % (tag, name)))
        out.append((plain, rich))
    launcher.barrier()
    if world > 1:
        if rank == 0:
            print('sample: %d requests on %d ranks, pipeline broadcast %.3f s (%d collectives)' % (len(reqs), world, bcast_s, launcher.LAST_BROADCAST_CALLS), flush=True)
        torch.distributed.destroy_process_group()
    return out[0] if len(out) == 1 else out


class _StandInArena:
    """CPU stand-in for an Engine / VaeDecoder in the dry launch: a byte arena, `arena_mark_bound`, `synchronize`."""

    def __init__(self, nbytes, fill):
        self.t = torch.full((nbytes,), fill, dtype=torch.uint8) if fill is not None else torch.zeros(nbytes, dtype=torch.uint8)
        self.bound = fill is not None
        self.device = "cpu"

    def arena_as_tensor(self):
        return self.t

    def arena_mark_bound(self):
        self.bound = True

    def synchronize(self):
        pass


class _StandInEncoder:
    def __init__(self, filled):
        g = torch.Generator().manual_seed(7)
        shapes = [((64, 32), torch.float32), ((32,), torch.float32), ((48, 32), torch.bfloat16), ((5,), torch.float32)]
        self.p = [(torch.randn(*s, generator=g).to(dt) if filled else torch.zeros(*s, dtype=dt)) for s, dt in shapes]

    def parameter_tensors(self):
        return self.p


def _dry_launch(a, rank, world, reqs, mine):
    """The control path of a `--gpus N` run without GPUs or checkpoints: the SAME launcher.broadcast_pipeline call build_model makes,
    on stand-in arenas (rank 0 filled, the others zero), then the per-rank request list as one JSON line."""
    from . import launcher
    unet, vae = _StandInArena(1 << 16, 0xA5 if rank == 0 else None), _StandInArena(1 << 12, 0x3C if rank == 0 else None)
    encs = [_StandInEncoder(rank == 0), _StandInEncoder(rank == 0)]
    seconds = launcher.broadcast_pipeline(unet, vae, encs, src=0)
    ref = _StandInEncoder(True)
    ok = bool((unet.t == 0xA5).all() and (vae.t == 0x3C).all() and unet.bound and vae.bound and
              all(torch.equal(x, y) for e in encs for x, y in zip(e.p, ref.p)))
    # ONE write per rank (line + newline together): the ranks share the launcher's stdout pipe, and print()'s separate newline write let two
    # ranks' lines run into each other once in ~25 runs
    sys.stdout.write(json.dumps({"rank": rank, "world": world, "requests_total": len(reqs), "requests_mine": [r["index"] for r in mine],
                                 "seeds_mine": [r["seed"] for r in mine], "pipeline_received": ok, "broadcast_collectives": launcher.LAST_BROADCAST_CALLS,
                                 "broadcast_s": seconds}) + "\n")
    sys.stdout.flush()
    launcher.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()
    if not ok:
        raise SystemExit("sample: dry launch: pipeline broadcast payload mismatch")
    return None


if __name__ == '__main__':
    main()
