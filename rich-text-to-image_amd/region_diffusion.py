"""Drop-in for `models/region_diffusion.py:RegionDiffusion` (SD-v1.5, PNDM/PLMS) with the denoising hot path
on the HIP engine.  Same method names / argument meaning / attributes as the reference."""
import torch

from .engine import SD15_CONFIG
from .schedulers import PNDMTables
from .unet import HipUNet2DConditionModel


class RegionDiffusion:
    def __init__(self, device=0, unet_state_dict=None, config=None, vae=None, tokenizer=None, text_encoder=None, load_path=None,
                 latent_hw=None):
        """`RegionDiffusion(device)` as sample.py:26-27 calls it: the reference loads runwayml/stable-diffusion-v1-5 there
        (rd.py:26-33); here the same id is resolved to a local diffusers-layout directory (checkpoint.resolve_checkpoint:
        `load_path` directory / $RTDIFF_SD_PATH / the Hugging Face hub cache) and UNet, VAE decoder, tokenizer and text encoder are
        loaded from it.  Callers that hold the weights already pass `unet_state_dict` (reference key names) and, optionally, VAE /
        CLIP objects with the diffusers / transformers call surface (`.decode(z).sample`, tokenizer(...), text_encoder(ids)[0])."""
        self.device_index = device if isinstance(device, int) else (torch.device(device).index or 0)
        self.device = torch.device(f"cuda:{self.device_index}")
        self.num_train_timesteps = 1000
        if unet_state_dict is None:
            from .checkpoint import load_components, resolve_checkpoint
            comp = load_components(resolve_checkpoint(load_path, "SD"), "SD", self.device_index, latent_hw)
            unet_state_dict, config = comp["unet_state_dict"], config or comp["config"]
            vae, tokenizer, text_encoder = vae or comp["vae"], tokenizer or comp["tokenizer"], text_encoder or comp["text_encoder"]
        self.vae, self.tokenizer, self.text_encoder = vae, tokenizer, text_encoder
        self.unet = HipUNet2DConditionModel(config or SD15_CONFIG, unet_state_dict, self.device_index)
        self.scheduler = PNDMTables(self.num_train_timesteps)          # rd.py:35-36
        self.alphas_cumprod = torch.tensor(self.scheduler.alphas_cumprod)
        self.masks = []
        self.attention_maps = None
        self.selfattn_maps = None
        self.crossattn_maps = None
        self.n_maps = None
        self.color_loss = torch.nn.functional.mse_loss

    # rd.py:49-84
    def get_text_embeds(self, prompt, negative_prompt):
        if self.tokenizer is None or self.text_encoder is None:
            raise RuntimeError("RegionDiffusion.get_text_embeds needs a CLIP tokenizer + text encoder (not available offline)")
        ti = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                            return_tensors="pt")
        with torch.no_grad():
            te = self.text_encoder(ti.input_ids.to(self.device))[0]
        ui = self.tokenizer(negative_prompt, padding="max_length", max_length=self.tokenizer.model_max_length, return_tensors="pt")
        with torch.no_grad():
            ue = self.text_encoder(ui.input_ids.to(self.device))[0]
        return torch.cat([ue, te])

    # rd.py:72-84
    def get_text_embeds_list(self, prompts):
        if self.tokenizer is None or self.text_encoder is None:
            raise RuntimeError("RegionDiffusion.get_text_embeds_list needs a CLIP tokenizer + text encoder (not available offline)")
        out = []
        for prompt in prompts:
            ti = self.tokenizer([prompt], padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                return_tensors="pt")
            with torch.no_grad():
                out.append(self.text_encoder(ti.input_ids.to(self.device))[0])
        return out

    # rd.py:238-246 - the VAE *encoder* is not on the rich-text path (nothing in the reference calls encode_imgs; only the
    # decoder is built here): callers that pass a VAE object with `.encode` still get the reference behaviour
    def encode_imgs(self, imgs):
        if not hasattr(self.vae, "encode"):
            raise NotImplementedError("encode_imgs needs a VAE with an encoder (diffusers AutoencoderKL call surface); the engine's "
                                      "VaeDecoder covers decode + colour guidance only - encode_imgs is unused by the rich-text flow")
        imgs = 2 * imgs - 1
        return self.vae.encode(imgs).latent_dist.sample() * 0.18215

    # rd.py:86-174
    def produce_latents(self, text_embeddings, height=512, width=512, num_inference_steps=50, guidance_scale=7.5,
                        latents=None, use_guidance=False, text_format_dict={}, inject_selfattn=0, inject_background=0,
                        elide_dead_forwards=False):
        if latents is None:
            latents = torch.randn((1, self.unet.in_channels, height // 8, width // 8), device=self.device)
        if use_guidance and not hasattr(self.vae, "color_guidance"):
            raise RuntimeError("use_guidance=True needs a rich_text_to_image_amd.engine.VaeDecoder as `vae` (rd.py:151-168)")
        n_styles = text_embeddings.shape[0] - 1
        assert n_styles == len(self.masks)                                  # rd.py:97
        h, w = latents.shape[2], latents.shape[3]
        n_prompts = text_embeddings.shape[0]
        eng = self.unet.engine(h, w, streams=n_prompts + 2, prompts=n_prompts)      # R+1 forwards, +2 reference forwards
        self.scheduler.set_timesteps(num_inference_steps)
        eng.set_prompts(text_embeddings.to(self.device))
        eng.set_masks([m.to(self.device) for m in self.masks])
        tfd = text_format_dict or {}
        eng.set_fontsize(tfd.get("word_pos"), tfd.get("font_size"))
        eng.set_schedule(1, self.scheduler.timesteps.tolist(), self.scheduler.table(), num_inference_steps)
        eng.set_latents(latents.to(self.device))
        for i, t in enumerate(self.scheduler.timesteps):
            if getattr(self, "split_image", False):      # intra-image split over the ranks of the process group (launcher.split_region_step)
                from .launcher import assert_ranks_agree, split_region_step
                if i % 10 == 0:                          # every rank must hold the same masks / latents (homogeneous ranks): fail loudly otherwise
                    if i == 0:
                        assert_ranks_agree(torch.cat([m.reshape(-1).float().cpu() for m in self.masks]), "the region masks")
                    assert_ranks_agree(eng.read_latents(latents.shape[-2], latents.shape[-1]), f"the latents before step {i}")
                split_region_step(eng, i, guidance_scale, inject_selfattn, inject_background, False, elide=elide_dead_forwards, defer_blend=use_guidance)
            else:
                eng.region_step(i, guidance_scale, inject_selfattn, inject_background, xl=False, elide=elide_dead_forwards,
                                defer_blend=use_guidance)
            if use_guidance:
                if t < tfd['guidance_start_step']:                           # rd.py:151
                    lat_ptr, eps_ptr = eng.state_ptrs()

                    def guide(lat_ptr=lat_ptr, eps_ptr=eps_ptr, t=t):
                        self.vae.color_guidance(lat_ptr, eps_ptr, float(self.scheduler.alphas_cumprod[int(t)]), h, w, tfd['color_obj_atten'],
                                                tfd['target_RGB'], tfd['color_guidance_weight'], tfd['color_obj_atten_all'])
                    if getattr(self, "split_image", False):              # rank 0 runs the VAE pass, the others receive the updated latents
                        from .launcher import guidance_from_rank0
                        guidance_from_rank0(eng, guide, h, w)
                    else:
                        guide()
                eng.background_blend()
        return eng.read_latents(h, w)

    def predict_x0(self, x_t, eps_t, t):                                    # rd.py:176-178
        a = self.alphas_cumprod[int(t)].to(x_t.device)
        return (x_t - eps_t * torch.sqrt(1 - a)) / torch.sqrt(a)

    # rd.py:180-225 (plain pass; attention-map capture = SURVEY 8a row a10, next)
    def produce_attn_maps(self, prompts, negative_prompts='', height=512, width=512, num_inference_steps=50,
                          guidance_scale=7.5, latents=None):
        if isinstance(prompts, str):
            prompts = [prompts]
        if isinstance(negative_prompts, str):
            negative_prompts = [negative_prompts]
        emb = self.get_text_embeds(prompts, negative_prompts)
        lat = self.plain_latents(emb, height, width, num_inference_steps, guidance_scale, latents)
        return self.latents_to_uint8(lat)

    def plain_latents(self, text_embeddings, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, latents=None):
        if latents is None:
            latents = torch.randn((1, self.unet.in_channels, height // 8, width // 8), device=self.device)
        h, w = latents.shape[2], latents.shape[3]
        n_prompts = text_embeddings.shape[0]
        eng = self.unet.engine(h, w, streams=n_prompts + 2, prompts=n_prompts)      # R+1 forwards, +2 reference forwards
        self.scheduler.set_timesteps(num_inference_steps)
        eng.set_prompts(text_embeddings.to(self.device))
        eng.set_schedule(1, self.scheduler.timesteps.tolist(), self.scheduler.table(), num_inference_steps)
        eng.set_latents(latents.to(self.device))
        hooks = getattr(self, "_tokenmap_hooks", False)
        if hooks:
            self._store_begin(eng)
        for i in range(len(self.scheduler.timesteps)):
            if getattr(self, "split_image", False):                      # one stream per rank, the text stream's rank records the maps
                from .launcher import split_plain_step
                split_plain_step(eng, i, guidance_scale)
            else:
                eng.plain_step(i, guidance_scale)
        if hooks:
            self._store_end(eng, len(self.scheduler.timesteps))
        return eng.read_latents(h, w)

    def decode_latents(self, latents):                                      # rd.py:227-236
        if self.vae is None:
            raise RuntimeError("no VAE bound: pass `vae=` (engine.VaeDecoder or an object with .decode(z).sample)")
        latents = 1 / 0.18215 * latents
        with torch.no_grad():
            imgs = self.vae.decode(latents)
            imgs = getattr(imgs, "sample", imgs)               # diffusers-style object or the engine's VaeDecoder tensor
        return (imgs / 2 + 0.5).clamp(0, 1)

    def latents_to_uint8(self, latents):
        imgs = self.decode_latents(latents)
        imgs = imgs.detach().cpu().permute(0, 2, 3, 1).numpy()
        return (imgs * 255).round().astype('uint8')

    # rd.py:248-273
    def prompt_to_img(self, prompts, negative_prompts='', height=512, width=512, num_inference_steps=50, guidance_scale=7.5,
                      latents=None, text_format_dict={}, use_guidance=False, inject_selfattn=0, inject_background=0):
        if isinstance(prompts, str):
            prompts = [prompts]
        if isinstance(negative_prompts, str):
            negative_prompts = [negative_prompts]
        text_embeds = self.get_text_embeds(prompts, negative_prompts)
        latents = self.produce_latents(text_embeds, height=height, width=width, latents=latents,
                                       num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                                       use_guidance=use_guidance, text_format_dict=text_format_dict,
                                       inject_selfattn=inject_selfattn, inject_background=inject_background)
        return self.latents_to_uint8(latents)

    # hook surface of the reference (rd.py:397-443): token-map capture is the "next" row f1
    def reset_attention_maps(self):                                         # rd.py:275-283
        for maps in (self.selfattn_maps, self.crossattn_maps):
            for key in (maps or {}):
                maps[key] = []

    def register_tokenmap_hooks(self):
        """rd.py:397-443: record head-averaged maps of the conditional half during produce_attn_maps / plain_latents.
        Recording happens on the GPU (rt_attn_store_*); the dicts are filled after each plain pass."""
        import collections
        self._tokenmap_hooks = True
        self.selfattn_maps = collections.defaultdict(list)
        self.crossattn_maps = collections.defaultdict(list)
        self.n_maps = collections.defaultdict(list)

    def remove_tokenmap_hooks(self):
        self._tokenmap_hooks = False
        self.selfattn_maps = self.crossattn_maps = self.n_maps = None

    def _store_begin(self, eng):
        from .attention_utils import CrossAttentionLayers, SelfAttentionLayers
        self._recorded = []
        for name, max_tokens, _ in eng.attn_modules():
            if name in SelfAttentionLayers and max_tokens <= 1024:
                eng.attn_store_enable(name, 2)       # rd.py:423: tests `name in crossattn_maps` => overwritten each step
                self._recorded.append(name)
            elif name in CrossAttentionLayers:
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
            if name.endswith("attn2") and name in tgt and not isinstance(tgt[name], list):
                tgt[name] = tgt[name] + m                # (on the GPU: get_token_maps averages there)
            else:
                tgt[name] = m
            eng.attn_store_enable(name, 0)
