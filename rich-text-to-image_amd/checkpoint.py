"""Checkpoint / text-side plumbing around the engine: diffusers-layout directory -> façade objects.

Replaces `StableDiffusionPipeline / AutoencoderKL / CLIPTextModel.from_pretrained` at rd.py:26-33 and xl.py:95-130.
The UNet and VAE decoder go to the HIP engine; the CLIP text encoders (once per prompt set, off the hot path) run on the same
operators through `clip_text_encoder.HipCLIPTextEncoder` (checked against transformers' CLIPTextModel on random weights).  No
real checkpoint exists offline: the loaders are exercised on synthetic diffusers-layout directories (tests/test_checkpoint_gpu.py).
"""
import glob
import json
import os

import torch

from .clip_tokenizer import ClipBPETokenizer


def load_state_dict_dir(path, prefer=("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                                      "model.fp16.safetensors")):
    """All tensors of the (possibly sharded) safetensors / .bin weights in one component directory."""
    from safetensors.torch import load_file
    for name in prefer:
        f = os.path.join(path, name)
        if os.path.exists(f):
            return load_file(f)
    shards = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if shards:
        sd = {}
        for f in shards:
            sd.update(load_file(f))
        return sd
    bins = sorted(glob.glob(os.path.join(path, "*.bin")))
    if bins:
        sd = {}
        for f in bins:
            sd.update(torch.load(f, map_location="cpu", weights_only=True))
        return sd
    raise FileNotFoundError(f"no weights under {path}")


def _component_config(path):
    f = os.path.join(path, "config.json")
    return json.load(open(f)) if os.path.exists(f) else {}


class ClipEncoderSD:
    """rd.py:49-84: `text_encoder(ids)[0]` = last hidden state (with final layer norm)."""

    def __init__(self, text_encoder, device):
        self.enc = text_encoder.to(device).eval()
        self.device = device

    def __call__(self, input_ids):
        with torch.no_grad():
            return (self.enc(input_ids.to(self.device))[0].float(),)


class ClipEncodersXL:
    """xl.py:318-440 for the cases the rich-text flow uses (num_images_per_prompt=1, CFG on, no LoRA / textual inversion):
    penultimate hidden states of both encoders concatenated on channels; pooled = projected embedding of encoder 2;
    negative prompt `None` + force_zeros_for_empty_prompt => zeros, otherwise encoded at the same max_length."""

    def __init__(self, tokenizers, text_encoders, device, force_zeros_for_empty_prompt=True):
        self.tokenizers = tokenizers
        self.encoders = [e.to(device).eval() for e in text_encoders]
        self.device = device
        self.force_zeros = force_zeros_for_empty_prompt

    def _encode(self, texts, max_length=None):
        hidden, pooled = [], None
        for tok, enc in zip(self.tokenizers, self.encoders):
            ids = tok(texts, padding="max_length", max_length=max_length or tok.model_max_length, truncation=True, return_tensors="pt").input_ids
            with torch.no_grad():
                out = enc(ids.to(self.device), output_hidden_states=True)
            pooled = out[0]                                     # the LAST encoder's wins (xl.py:353)
            hidden.append(out.hidden_states[-2])
        return torch.cat(hidden, dim=-1).float(), pooled.float()

    def __call__(self, prompt, negative_prompt):
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        pe, pp = self._encode(prompt)
        if negative_prompt is None and self.force_zeros:
            return pe, torch.zeros_like(pe), pp, torch.zeros_like(pp)
        neg = negative_prompt or ""
        neg = [neg] if isinstance(neg, str) else list(neg)
        ne, npool = self._encode(neg, max_length=pe.shape[1])
        return pe, ne, pp, npool


_UNET_KEYS = ("in_channels", "out_channels", "block_out_channels", "down_block_types", "up_block_types", "layers_per_block",
              "transformer_layers_per_block", "attention_head_dim", "cross_attention_dim", "norm_num_groups", "norm_eps", "use_linear_projection",
              "addition_embed_type", "addition_time_embed_dim", "projection_class_embeddings_input_dim")


def unet_config(path, default):
    """`unet/config.json` (UNet2DConditionModel kwargs) over the defaults of the model family; absent keys keep diffusers' defaults."""
    cfg = dict(default)
    js = _component_config(path)
    # options of UNet2DConditionModel that the engine does not implement (none of them is used by SD-v1.5 / SDXL-base): fail loudly
    unsupported = {"resnet_time_scale_shift": ("default",), "class_embed_type": (None,), "dual_cross_attention": (False,),
                   "only_cross_attention": (False,), "encoder_hid_dim": (None,), "time_embedding_type": ("positional",),
                   "mid_block_type": ("UNetMidBlock2DCrossAttn",), "act_fn": ("silu",), "conv_in_kernel": (3,), "conv_out_kernel": (3,)}
    for k, ok in unsupported.items():
        if k in js and js[k] not in ok:
            raise ValueError(f"unet/config.json: {k}={js[k]!r} is not supported by the engine (supported: {ok[0]!r})")
    if js:
        cfg.update({"transformer_layers_per_block": 1, "use_linear_projection": False, "norm_eps": 1e-5, "addition_embed_type": None,
                    "addition_time_embed_dim": None, "projection_class_embeddings_input_dim": None})
        cfg.update({k: js[k] for k in _UNET_KEYS if k in js and js[k] is not None})
        for k in ("block_out_channels", "down_block_types", "up_block_types"):
            cfg[k] = tuple(cfg[k])
        for k in ("transformer_layers_per_block", "attention_head_dim", "layers_per_block"):
            if isinstance(cfg[k], list):
                cfg[k] = tuple(cfg[k])
    return cfg


def vae_config(path, default):
    cfg = dict(default)
    js = _component_config(path)
    for k in ("block_out_channels", "layers_per_block", "norm_num_groups", "scaling_factor", "latent_channels", "out_channels"):
        if js.get(k) is not None:
            cfg[k] = tuple(js[k]) if isinstance(js[k], list) else js[k]
    return cfg


def load_text_encoder(path, device=0, with_projection=False, weights=True):
    """CLIP text encoder directory (config.json + safetensors) -> HipCLIPTextEncoder (GEMMs / LayerNorm / causal attention on the
    engine's operators; no transformers model is instantiated).  weights=False: the same object from config.json alone with
    zero-filled parameters of the right shapes - a rank that receives them through launcher.broadcast_pipeline."""
    from .clip_text_encoder import HipCLIPTextEncoder, empty_state_dict
    cfg = _component_config(path)
    sd = load_state_dict_dir(path) if weights else empty_state_dict(cfg, with_projection)
    return HipCLIPTextEncoder(sd, cfg, device=device, with_projection=with_projection)


DEFAULT_REPO = {"SD": "runwayml/stable-diffusion-v1-5", "SDXL": "stabilityai/stable-diffusion-xl-base-1.0"}
ENV_OVERRIDE = {"SD": "RTDIFF_SD_PATH", "SDXL": "RTDIFF_SDXL_PATH"}


def resolve_checkpoint(name_or_path, kind="SD"):
    """What `from_pretrained(name_or_path)` resolves at rd.py:26-33 / xl.py:105-120, without a network: a diffusers-layout directory.
    Order: (1) `name_or_path` is a directory; (2) the environment override of the family (RTDIFF_SD_PATH / RTDIFF_SDXL_PATH) when
    the name is the family's default hub id or None; (3) the Hugging Face hub cache (`$HUGGINGFACE_HUB_CACHE`, `$HF_HOME/hub`,
    `~/.cache/huggingface/hub`): models--<org>--<name>/snapshots/<rev>/ - what a previous `from_pretrained` download left behind.
    Raises FileNotFoundError naming everything that was tried (there is no silent fallback to random weights)."""
    tried = []
    name = name_or_path or DEFAULT_REPO[kind]
    if os.path.isdir(name) and os.path.isdir(os.path.join(name, "unet")):
        return name
    tried.append(name)
    if name == DEFAULT_REPO[kind] and os.environ.get(ENV_OVERRIDE[kind]):
        p = os.environ[ENV_OVERRIDE[kind]]
        if os.path.isdir(os.path.join(p, "unet")):
            return p
        tried.append(f"${ENV_OVERRIDE[kind]}={p}")
    if "/" in name and not os.path.isabs(name):
        roots = [os.environ.get("HUGGINGFACE_HUB_CACHE"), os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
                 os.path.expanduser("~/.cache/huggingface/hub")]
        for root in [r for r in roots if r]:
            snaps = sorted(glob.glob(os.path.join(root, "models--" + name.replace("/", "--"), "snapshots", "*")), key=os.path.getmtime)
            for snap in reversed(snaps):
                if os.path.isdir(os.path.join(snap, "unet")):
                    return snap
            tried.append(os.path.join(root, "models--" + name.replace("/", "--")))
    raise FileNotFoundError(f"no diffusers-layout checkpoint for {name!r} ({kind}); tried: " + "; ".join(tried) +
                            f". Pass a local directory (unet/, vae/, tokenizer*/, text_encoder*/) or set ${ENV_OVERRIDE[kind]}.")


def load_guidance_vae(load_path, kind, device=0, latent_hw=None):
    """A ONE-PASS bf16 VaeDecoder on the checkpoint's VAE weights, for `RegionDiffusionXL.guidance_vae` (sample.py --guidance_precision bf16):
    the colour-guidance pass alone in one bf16 MFMA pass while the final decode stays on the precise engine."""
    from .engine import SD_VAE_CONFIG, SDXL_VAE_CONFIG, VaeDecoder
    vae_sd = load_state_dict_dir(os.path.join(load_path, "vae"))
    vae_cfg = vae_config(os.path.join(load_path, "vae"), SD_VAE_CONFIG if kind == "SD" else SDXL_VAE_CONFIG)
    hw = latent_hw or ((64, 64) if kind == "SD" else (128, 128))
    return VaeDecoder(vae_cfg, hw[0], hw[1], device=device, state_dict=vae_sd, precise=False)


def load_components(load_path, kind="SD", device=0, latent_hw=None, lora_path=None, lora_scale=1.0, weights=True):
    """Everything the facade constructors need from a diffusers-layout directory (unet/, vae/, tokenizer[_2]/, text_encoder[_2]/) as
    keyword arguments of RegionDiffusion / RegionDiffusionXL.  `latent_hw` sizes the VAE plan (default: the model's native size).
    `lora_path`: a LoRA .safetensors file merged into the UNet weights before they are bound (lora.merge_lora).
    weights=False (seed-parallel ranks != 0, sample.py --gpus N): only the config / tokenizer files are read; the UNet arena, the VAE
    arena and the text-encoder parameters stay zero until launcher.broadcast_pipeline delivers rank 0's."""
    from .engine import SD15_CONFIG, SD_VAE_CONFIG, SDXL_CONFIG, SDXL_VAE_CONFIG, VaeDecoder
    dev = torch.device(f"cuda:{device}")
    unet_sd = load_state_dict_dir(os.path.join(load_path, "unet")) if weights else "empty"
    vae_sd = load_state_dict_dir(os.path.join(load_path, "vae")) if weights else None
    if lora_path and weights:
        from safetensors.torch import load_file
        from .lora import merge_lora
        unet_sd, _ = merge_lora(unet_sd, load_file(lora_path), lora_scale)
    unet_cfg = unet_config(os.path.join(load_path, "unet"), SD15_CONFIG if kind == "SD" else SDXL_CONFIG)
    vae_cfg = vae_config(os.path.join(load_path, "vae"), SD_VAE_CONFIG if kind == "SD" else SDXL_VAE_CONFIG)
    hw = latent_hw or ((64, 64) if kind == "SD" else (128, 128))
    vae = VaeDecoder(vae_cfg, hw[0], hw[1], device=device, state_dict=vae_sd, precise=(kind == "SDXL"))   # xl.py:856 / :918-938: fp32 VAE
    tok = ClipBPETokenizer.from_pretrained(load_path, "tokenizer")
    if kind == "SD":
        enc = load_text_encoder(os.path.join(load_path, "text_encoder"), device, weights=weights)
        return dict(unet_state_dict=unet_sd, config=unet_cfg, vae=vae, tokenizer=tok, text_encoder=ClipEncoderSD(enc, dev))
    tok2 = ClipBPETokenizer.from_pretrained(load_path, "tokenizer_2")
    enc1 = load_text_encoder(os.path.join(load_path, "text_encoder"), device, weights=weights)
    enc2 = load_text_encoder(os.path.join(load_path, "text_encoder_2"), device, with_projection=True, weights=weights)
    fz = True
    mi = os.path.join(load_path, "model_index.json")
    if os.path.exists(mi):
        fz = json.load(open(mi)).get("force_zeros_for_empty_prompt", True)
    return dict(unet_state_dict=unet_sd, config=unet_cfg, vae=vae, tokenizer=tok, vae_scaling_factor=vae_cfg["scaling_factor"],
                text_encoders=ClipEncodersXL([tok, tok2], [enc1, enc2], dev, fz))


def load_pipeline(load_path, kind="SD", device=0, latent_hw=None, lora_path=None, lora_scale=1.0):
    """kind 'SD' -> RegionDiffusion, 'SDXL' -> RegionDiffusionXL, from a diffusers-layout directory or hub id (resolve_checkpoint)."""
    comp = load_components(resolve_checkpoint(load_path, kind), kind, device, latent_hw, lora_path, lora_scale)
    if kind == "SD":
        from .region_diffusion import RegionDiffusion
        return RegionDiffusion(device, **comp)
    from .region_diffusion_sdxl import RegionDiffusionXL
    return RegionDiffusionXL(load_path, device, **comp)
