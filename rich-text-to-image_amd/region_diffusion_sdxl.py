"""Drop-in for `models/region_diffusion_sdxl.py:RegionDiffusionXL` (SDXL, Euler) with the denoising hot path on
the HIP engine.  `sample()` keeps the reference's signature (xl.py:556-587)."""
import types

import torch

from .engine import SDXL_CONFIG
from .schedulers import EulerTables
from .unet import HipUNet2DConditionModel


class StableDiffusionXLPipelineOutput(dict):
    def __init__(self, images):
        super().__init__(images=images)
        self.images = images


class RegionDiffusionXL:
    def __init__(self, load_path=None, device=0, unet_state_dict=None, config=None, vae=None, text_encoders=None,
                 vae_scaling_factor=0.13025, tokenizer=None, latent_hw=None):
        """`RegionDiffusionXL(load_path="stabilityai/stable-diffusion-xl-base-1.0")` as sample.py:28-30 calls it (xl.py:105-120
        loads every component from `load_path`): a diffusers-layout directory, or a hub id resolved to one without a network
        (checkpoint.resolve_checkpoint: $RTDIFF_SDXL_PATH for the default id, then the Hugging Face hub cache).  Callers that hold
        the weights pass `unet_state_dict` (+ optional vae / text_encoders / tokenizer) and `load_path` is not read."""
        self.device_index = device if isinstance(device, int) else (torch.device(device).index or 0)
        self.device = torch.device(f"cuda:{self.device_index}")
        self.device_type = "cuda"
        if unet_state_dict is None:
            from .checkpoint import load_components, resolve_checkpoint
            comp = load_components(resolve_checkpoint(load_path, "SDXL"), "SDXL", self.device_index, latent_hw)
            unet_state_dict, config = comp["unet_state_dict"], config or comp["config"]
            vae, text_encoders, tokenizer = vae or comp["vae"], text_encoders or comp["text_encoders"], tokenizer or comp["tokenizer"]
            vae_scaling_factor = comp["vae_scaling_factor"]
        self.unet = HipUNet2DConditionModel(config or SDXL_CONFIG, unet_state_dict, self.device_index)
        self.vae = vae
        # Optional (round 6): a second VaeDecoder used ONLY for the colour-guidance pass (xl.py:849-867).  The reference runs the SDXL VAE in
        # fp32 because it overflows in fp16 (xl.py:856), and `vae` (precise = three bf16 MFMA passes) follows it.  A one-pass bf16 engine has
        # fp32's range; over the 50-step guided schedule it leaves the latents where the precise engine leaves them (update-relative error
        # against the fp32 oracle 3.05e-2 -> 1.16e-2 either way, LABNOTES R6.6) for 37 instead of 80 ms per step.  None: `vae` guides too.
        self.guidance_vae = None
        self.text_encoders = text_encoders
        self.vae_scaling_factor = vae_scaling_factor
        self.vae_scale_factor = 8
        self.default_sample_size = 128
        self.scheduler = EulerTables()
        self.masks = []
        self.selfattn_maps = self.crossattn_maps = self.n_maps = None
        self.attention_maps = None                                   # xl.py:132 (only the evaluation hooks ever set it)
        self.tokenizer = tokenizer                                   # richtext_utils needs `model.tokenizer._tokenize`

    def reset_attention_maps(self):
        for maps in (self.selfattn_maps, self.crossattn_maps):
            for key in (maps or {}):
                maps[key] = []

    def check_inputs(self, prompt, height, width, prompt_embeds, pooled_prompt_embeds):          # xl.py:462-519
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.")

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size):                # xl.py:539-553
        return torch.tensor([list(original_size + crops_coords_top_left + target_size)], dtype=torch.float32)

    def encode_prompt(self, prompt, negative_prompt):
        if self.text_encoders is None:
            raise RuntimeError("RegionDiffusionXL needs CLIP text encoders for string prompts (not available offline); "
                               "pass prompt_embeds / pooled_prompt_embeds instead")
        return self.text_encoders(prompt, negative_prompt)

    def sample(self, prompt=None, prompt_2=None, height=None, width=None, num_inference_steps=50, guidance_scale=5.0,
               negative_prompt=None, negative_prompt_2=None, num_images_per_prompt=1, eta=0.0, generator=None,
               latents=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
               negative_pooled_prompt_embeds=None, output_type="pil", return_dict=True, callback=None, callback_steps=1,
               cross_attention_kwargs=None, guidance_rescale=0.0, original_size=None, crops_coords_top_left=(0, 0),
               target_size=None, use_guidance=False, inject_selfattn=0, inject_background=0, text_format_dict=None,
               run_rich_text=False, elide_dead_forwards=False):
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        self.check_inputs(prompt, height, width, prompt_embeds, pooled_prompt_embeds)
        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = \
                self.encode_prompt(prompt, negative_prompt)
        do_cfg = guidance_scale > 1.0
        if run_rich_text and do_cfg and guidance_rescale > 0.0:
            raise NotImplementedError                                                              # xl.py:827-830
        if use_guidance and not hasattr(self.vae, "color_guidance"):
            raise RuntimeError("use_guidance=True needs a rich_text_to_image_amd.engine.VaeDecoder as `vae` (xl.py:849-867)")
        self.scheduler.set_timesteps(num_inference_steps)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        if latents is None:
            latents = torch.randn((1, 4, h, w), generator=generator, device=self.device if generator is None else generator.device)
        latents = latents.to(self.device).float() * self.scheduler.init_noise_sigma                # xl.py:533-536
        add_time_ids = self._get_add_time_ids(tuple(original_size), tuple(crops_coords_top_left), tuple(target_size))
        embeds = torch.cat([negative_prompt_embeds, prompt_embeds], 0).to(self.device).float()    # xl.py:760
        pooled = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], 0).to(self.device).float()
        eng = self.unet.engine(h, w, streams=embeds.shape[0] + 2 if run_rich_text else 2, prompts=embeds.shape[0])
        eng.set_prompts(embeds, pooled, add_time_ids)
        eng.set_schedule(0, self.scheduler.timesteps.tolist(), self.scheduler.table(), num_inference_steps)
        eng.set_latents(latents)
        n = len(self.scheduler.timesteps)
        if run_rich_text:
            n_styles = embeds.shape[0] - 1
            assert n_styles == len(self.masks), (n_styles, len(self.masks))
            eng.set_masks([m.to(self.device) for m in self.masks])
            tfd = text_format_dict or {}
            eng.set_fontsize(tfd.get("word_pos"), tfd.get("font_size"))
            for i in range(n):
                if getattr(self, "split_image", False):
                    # intra-image split (launcher.split_region_step): the ranks of the process group share THIS image's step - each runs
                    # the forwards of its stream range, one exchange of the noise predictions, every rank finishes the step
                    from .launcher import assert_ranks_agree, split_region_step
                    if i % 10 == 0:                      # every rank must hold the same masks / latents (homogeneous ranks): fail loudly otherwise
                        if i == 0:
                            assert_ranks_agree(torch.cat([m.reshape(-1).float().cpu() for m in self.masks]), "the region masks")
                        assert_ranks_agree(eng.read_latents(h, w), f"the latents before step {i}")
                    split_region_step(eng, i, guidance_scale, inject_selfattn, inject_background, True, elide=elide_dead_forwards, defer_blend=use_guidance)
                else:
                    eng.region_step(i, guidance_scale, inject_selfattn, inject_background, xl=True, elide=elide_dead_forwards,
                                    defer_blend=use_guidance)
                if use_guidance:
                    t = float(self.scheduler.timesteps[i])
                    if t < tfd['guidance_start_step']:                   # xl.py:849; predict_x0 on the unscaled Euler latents (quirk 4)
                        lat_ptr, eps_ptr = eng.state_ptrs()

                        def guide(lat_ptr=lat_ptr, eps_ptr=eps_ptr, t=t):
                            (self.guidance_vae or self.vae).color_guidance(lat_ptr, eps_ptr, float(self.scheduler.alphas_cumprod[int(t)]), h, w, tfd['color_obj_atten'],
                                                    tfd['target_RGB'], tfd['color_guidance_weight'], tfd['color_obj_atten_all'])
                        if getattr(self, "split_image", False):          # rank 0 runs the VAE pass, the others receive the updated latents
                            from .launcher import guidance_from_rank0
                            guidance_from_rank0(eng, guide, h, w)
                        else:
                            guide()
                    eng.background_blend()
                if callback is not None and i % callback_steps == 0:
                    callback(i, self.scheduler.timesteps[i], eng.read_latents(h, w))
        else:
            hooks = getattr(self, "_tokenmap_hooks", False)
            if hooks:
                self._store_begin(eng)
            for i in range(n):
                if getattr(self, "split_image", False):                  # one stream per rank, the text stream's rank records the maps
                    from .launcher import split_plain_step
                    split_plain_step(eng, i, guidance_scale)
                else:
                    eng.plain_step(i, guidance_scale)
            if hooks:
                self._store_end(eng, n)
        latents = eng.read_latents(h, w)
        if output_type == "latent":
            return StableDiffusionXLPipelineOutput(images=latents)
        if self.vae is None:
            raise RuntimeError("no VAE bound: pass `vae=` (engine.VaeDecoder or an object with .decode(z)) or use output_type='latent'")
        image = self.vae.decode(latents / self.vae_scaling_factor)
        image = getattr(image, "sample", image)
        if output_type == "pt":
            return StableDiffusionXLPipelineOutput(images=image)
        arr = ((image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy() * 255).round().astype("uint8")    # VaeImageProcessor.postprocess
        if output_type == "np":
            return StableDiffusionXLPipelineOutput(images=arr)
        from PIL import Image
        return StableDiffusionXLPipelineOutput(images=[Image.fromarray(a) for a in arr])

    def predict_x0(self, x_t, eps_t, t):                                                           # xl.py:955-957
        a = torch.tensor(self.scheduler.alphas_cumprod)[int(t)].to(x_t.device)
        return (x_t - eps_t * torch.sqrt(1 - a)) / torch.sqrt(a)

    cross_attention_layers = None      # defaults to attention_utils.CrossAttentionLayers_XL

    def register_tokenmap_hooks(self):
        """xl.py:959-1016: every attn1 map is accumulated by the reference; only the 32x32 ones are consumed
        (attention_utils.py:243-248), so only those are recorded here (GPU side, rt_attn_store_*)."""
        import collections
        self._tokenmap_hooks = True
        self.selfattn_maps = collections.defaultdict(list)
        self.crossattn_maps = collections.defaultdict(list)
        self.n_maps = collections.defaultdict(list)

    def remove_tokenmap_hooks(self):
        self._tokenmap_hooks = False
        self.selfattn_maps = self.crossattn_maps = self.n_maps = None

    def _store_begin(self, eng):
        from .attention_utils import CrossAttentionLayers_XL
        cross = self.cross_attention_layers or CrossAttentionLayers_XL
        self._recorded = []
        for name, max_tokens, _ in eng.attn_modules():
            if name.endswith("attn1") and max_tokens <= 1024:
                eng.attn_store_enable(name, 1)
                self._recorded.append(name)
            elif name in cross:
                eng.attn_store_enable(name, 1)
                self._recorded.append(name)
            else:
                eng.attn_store_enable(name, 0)
        eng.attn_store_reset()

    def _store_end(self, eng, n_calls):
        for name, _, _ in eng.attn_modules():
            self.n_maps[name] = (self.n_maps[name] if name in self.n_maps else 0) + n_calls
        for name in self._recorded:
            n, m = eng.attn_store_read(name)
            if m is None:
                continue
            tgt = self.crossattn_maps if name.endswith("attn2") else self.selfattn_maps
            # the maps stay on the GPU (60 x 4 MB at SDXL): get_token_maps averages them there and moves ONE 1024 x 1024 affinity to the host
            tgt[name] = (tgt[name] + m) if (name in tgt and not isinstance(tgt[name], list)) else m
            eng.attn_store_enable(name, 0)
