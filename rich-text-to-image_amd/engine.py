"""ctypes binding of librtdiff.so (include/rtdiff.h).  PyTorch is used for device memory only."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librtdiff.so")
_lib = None

RT_MAX_LEVELS = 4
RT_MAX_STREAMS = 16
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
SCHED_EULER, SCHED_PNDM = 0, 1
A_DENSE, A_CONV3, A_CONV3_S2, A_CONV3_UP2 = 0, 1, 2, 3
EPI_BF16, EPI_F32, EPI_BF16_TEMB, EPI_GEGLU, EPI_F16 = 0, 1, 2, 3, 4

# UNet architectures of the two pipelines (models/region_diffusion.py:32, region_diffusion_sdxl.py:115;
# values = the published unet/config.json of runwayml/stable-diffusion-v1-5 and
# stabilityai/stable-diffusion-xl-base-1.0, validated by parameter count in SURVEY.md section 8)
SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    layers_per_block=2, transformer_layers_per_block=1, attention_head_dim=8, cross_attention_dim=768,
    norm_num_groups=32, norm_eps=1e-5, use_linear_projection=False, addition_embed_type=None,
    addition_time_embed_dim=None, projection_class_embeddings_input_dim=None)
SDXL_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280),
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    layers_per_block=2, transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20),
    cross_attention_dim=2048, norm_num_groups=32, norm_eps=1e-5, use_linear_projection=True,
    addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816)


class RtConfig(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int),
        ("block_out_channels", C.c_int * RT_MAX_LEVELS),
        ("down_has_attn", C.c_int * RT_MAX_LEVELS),
        ("up_has_attn", C.c_int * RT_MAX_LEVELS),
        ("layers_per_block", C.c_int * RT_MAX_LEVELS),
        ("transformer_layers", C.c_int * RT_MAX_LEVELS),
        ("heads", C.c_int * RT_MAX_LEVELS),
        ("cross_attention_dim", C.c_int),
        ("norm_groups", C.c_int),
        ("norm_eps", C.c_float),
        ("use_linear_projection", C.c_int),
        ("addition_text_time", C.c_int),
        ("addition_time_embed_dim", C.c_int),
        ("projection_class_embeddings_input_dim", C.c_int),
        ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("latent_h", C.c_int), ("latent_w", C.c_int),
        ("max_streams", C.c_int), ("max_prompts", C.c_int),
    ]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librtdiff error {code}: {msg}")
        self.code = code


_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_synchronize", "rt_weight_count",
    "rt_weight_info", "rt_bind_weight", "rt_weights_missing", "rt_arena_info", "rt_arena_mark_bound",
    "rt_set_prompts", "rt_set_masks", "rt_set_fontsize", "rt_set_schedule", "rt_set_latents", "rt_get_latents",
    "rt_region_step", "rt_plain_step", "rt_unet_forward", "rt_op_gemm", "rt_op_attention", "rt_op_groupnorm",
    "rt_op_layernorm", "rt_op_layernorm_f16", "rt_op_small_linear", "rt_op_timestep_embed", "rt_op_last_error", "rt_profile_enable",
    "rt_profile_read", "rt_op_gemm_force_config", "rt_op_gemm_debug", "rt_attn_store_enable", "rt_attn_store_reset",
    "rt_attn_store_read", "rt_attn_module_count", "rt_attn_module_info", "rt_get_state_ptrs", "rt_background_blend", "rt_vae_create", "rt_vae_destroy",
    "rt_vae_last_error", "rt_vae_weight_count", "rt_vae_weight_info", "rt_vae_bind_weight", "rt_vae_synchronize", "rt_vae_decode",
    "rt_vae_color_guidance", "rt_vae_arena_info", "rt_vae_arena_mark_bound", "rt_op_cast_bf16", "rt_op_attention_probs_avg", "rt_op_embed", "rt_op_activation", "rt_op_causal_attention",
    "rt_op_cross_attn_block", "rt_op_gemm16_variant", "rt_op_gemm16_pick", "rt_op_split_plan", "rt_profile_read2", "rt_op_gemm_qk_vt", "rt_op_gemm_pair_pick",
    "rt_region_step_part", "rt_region_step_finish", "rt_eps_info", "rt_op_split_range",
    "rt_op_ln_gemm", "rt_op_gemm_emit_partials", "rt_op_ln_partials", "rt_op_probes_built", "rt_op_attention_units_plan",
    "rt_plain_step_part", "rt_plain_step_finish",
]


def load_library(path=None):
    """Loads librtdiff.so; fails loudly when the HIP extension has not been built (no CPU fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RTDIFF_LIB_PATH") or _LIB_PATH          # RTDIFF_LIB_PATH: A/B builds of the same ABI (benchmarks only)
    if not os.path.exists(p):
        raise RtError(-3, f"{p} not found: build it with `python __graft_entry__.py build` "
                          "(hipcc --offload-arch=gfx950); there is no fallback path")
    lib = C.CDLL(p)
    # RTDIFF_ALLOW_MISSING_SYMBOLS: A/B runs against an OLDER build of the library (benchmarks / regression hunts only)
    names = [s for s in _SYMBOLS if hasattr(lib, s)] if os.environ.get("RTDIFF_ALLOW_MISSING_SYMBOLS") else _SYMBOLS
    for s in names:
        getattr(lib, s)          # AttributeError if a declared symbol is missing
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_op_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [C.c_void_p]
    lib.rt_vae_last_error.restype = C.c_char_p
    lib.rt_vae_last_error.argtypes = [C.c_void_p]
    for name in names:
        if name not in ("rt_last_error", "rt_op_last_error", "rt_vae_last_error"):
            getattr(lib, name).restype = C.c_int
    flags = int(os.environ.get("RTDIFF_DEBUG_FLAGS", "0"))      # A/B switches of rt_op_gemm_debug (benchmarks only)
    if flags:
        lib.rt_op_gemm_debug(flags)
    if path is None:
        _lib = lib
    return lib


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def config_from_dict(cfg, latent_h, latent_w, max_streams=8, max_prompts=8):
    """UNet2DConditionModel kwargs (reference config.json) -> rt_config."""
    c = RtConfig()
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    c.n_levels = n
    lpb, tl, heads = _tup(cfg["layers_per_block"], n), _tup(cfg["transformer_layers_per_block"], n), _tup(cfg["attention_head_dim"], n)
    for i in range(n):
        c.block_out_channels[i] = boc[i]
        c.down_has_attn[i] = int(cfg["down_block_types"][i] == "CrossAttnDownBlock2D")
        c.up_has_attn[i] = int(cfg["up_block_types"][i] == "CrossAttnUpBlock2D")
        c.layers_per_block[i] = lpb[i]
        c.transformer_layers[i] = tl[i]
        c.heads[i] = heads[i]
        if cfg["down_block_types"][i] not in ("CrossAttnDownBlock2D", "DownBlock2D") or \
           cfg["up_block_types"][i] not in ("CrossAttnUpBlock2D", "UpBlock2D"):
            raise ValueError("unsupported block type (only the SD / SDXL block types are on the hot path)")
    c.cross_attention_dim = cfg["cross_attention_dim"]
    c.norm_groups = cfg["norm_num_groups"]
    c.norm_eps = cfg.get("norm_eps", 1e-5)
    c.use_linear_projection = int(bool(cfg["use_linear_projection"]))
    c.addition_text_time = int(cfg.get("addition_embed_type") == "text_time")
    c.addition_time_embed_dim = cfg.get("addition_time_embed_dim") or 0
    c.projection_class_embeddings_input_dim = cfg.get("projection_class_embeddings_input_dim") or 0
    c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
    c.latent_h, c.latent_w = latent_h, latent_w
    c.max_streams, c.max_prompts = max_streams, max_prompts
    return c


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """One engine = one GPU.  Mirrors the C ABI one to one; see include/rtdiff.h for semantics."""

    def __init__(self, cfg_dict, latent_h, latent_w, device=0, max_streams=8, max_prompts=8):
        self.lib = load_library()
        self.cfg_dict = dict(cfg_dict)
        self.cfg = config_from_dict(cfg_dict, latent_h, latent_w, max_streams, max_prompts)
        self.h = C.c_void_p()
        self.device = device
        rc = self.lib.rt_create(C.byref(self.cfg), C.c_int(device), C.byref(self.h))
        if rc != 0:
            raise RtError(rc, self.lib.rt_last_error(None).decode())
        self._keep = []

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_last_error(self.h).decode())

    # ---- weights
    def weight_table(self):
        n = self.lib.rt_weight_count(self.h)
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(n):
            self._chk(self.lib.rt_weight_info(self.h, i, name, 256, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def bind_weight(self, name, t):
        import torch
        dt = {torch.float32: DTYPE_F32, torch.float16: DTYPE_F16, torch.bfloat16: DTYPE_BF16}[t.dtype]
        t = t.contiguous()
        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
        self._chk(self.lib.rt_bind_weight(self.h, name.encode(), _ptr(t), dt, shape, t.dim()))

    def load_state_dict(self, sd, device=None):
        """Packs a reference-layout state_dict (CPU or GPU tensors) into the engine's bf16 arena."""
        import torch
        dev = device or f"cuda:{self.device}"
        for name, shape in self.weight_table():
            if name not in sd:
                raise RtError(-4, f"missing weight {name}")
            t = sd[name].to(dev, non_blocking=False)
            self.bind_weight(name, t)
        self.synchronize()

    def init_random_weights(self, seed=0, std_scale=1.0):
        """Random-init weights of the true architecture directly on the GPU (bench / smoke: no checkpoints offline)."""
        import math
        import torch
        dev = f"cuda:{self.device}"
        g = torch.Generator(device=dev).manual_seed(seed)
        for name, shape in self.weight_table():
            if name.endswith(".weight") and len(shape) >= 2:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                t = (torch.rand(shape, generator=g, device=dev) * 2 - 1) * (std_scale / math.sqrt(fan_in))
            elif name.endswith(".weight"):
                t = 1.0 + 0.1 * (torch.rand(shape, generator=g, device=dev) * 2 - 1)
            else:
                t = 0.05 * (torch.rand(shape, generator=g, device=dev) * 2 - 1)
            self.bind_weight(name, t)
            self.synchronize()
            del t

    def weights_missing(self):
        buf = C.create_string_buffer(1 << 16)
        n = self.lib.rt_weights_missing(self.h, buf, len(buf))
        return n, [s for s in buf.value.decode().split(";") if s]

    def arena(self):
        p, b = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.rt_arena_info(self.h, C.byref(p), C.byref(b)))
        return p.value, b.value

    def arena_mark_bound(self):
        self._chk(self.lib.rt_arena_mark_bound(self.h))

    def synchronize(self):
        self._chk(self.lib.rt_synchronize(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.rt_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- per image
    def set_prompts(self, prompt_embeds, pooled=None, time_ids=None):
        import torch
        pe = prompt_embeds.contiguous().float()
        assert pe.dim() == 3 and pe.shape[1] == 77
        pl = pooled.contiguous().float() if pooled is not None else None
        tid = (C.c_float * 6)(*[float(v) for v in time_ids.flatten().tolist()[:6]]) if time_ids is not None else None
        self._chk(self.lib.rt_set_prompts(self.h, _ptr(pe), _ptr(pl), tid, pe.shape[0], pl.shape[1] if pl is not None else 0))

    def set_masks(self, masks):
        """masks: list of [1,4,h,w] tensors (model.masks) or one [R,4,h,w] tensor."""
        import torch
        m = torch.cat(list(masks), 0) if isinstance(masks, (list, tuple)) else masks
        m = m.contiguous().float()
        self._chk(self.lib.rt_set_masks(self.h, _ptr(m), m.shape[0], m.shape[2], m.shape[3]))
        self.synchronize()

    def set_fontsize(self, word_pos=None, font_size=None):
        if word_pos is None or font_size is None or len(word_pos) == 0:
            self._chk(self.lib.rt_set_fontsize(self.h, None, None, 0))
            return
        wp = [int(v) for v in word_pos.tolist()]
        fs = [float(v) for v in font_size.tolist()]
        self._chk(self.lib.rt_set_fontsize(self.h, (C.c_int64 * len(wp))(*wp), (C.c_float * len(fs))(*fs), len(wp)))

    def set_schedule(self, kind, timesteps, table, num_inference_steps):
        ts = [float(v) for v in timesteps]
        tb = [float(v) for v in table]
        self._chk(self.lib.rt_set_schedule(self.h, kind, (C.c_float * len(ts))(*ts), len(ts), (C.c_float * len(tb))(*tb),
                                           len(tb), num_inference_steps))

    def set_latents(self, latents):
        l = latents.contiguous().float()
        assert l.shape[0] == 1 and l.shape[1] == 4
        self._chk(self.lib.rt_set_latents(self.h, _ptr(l), l.shape[2], l.shape[3]))
        self.synchronize()

    # ---- token-map attention store
    def attn_modules(self):
        n = self.lib.rt_attn_module_count(self.h)
        name = C.create_string_buffer(256)
        mt, hd = C.c_int(), C.c_int()
        out = []
        for i in range(n):
            self._chk(self.lib.rt_attn_module_info(self.h, i, name, 256, C.byref(mt), C.byref(hd)))
            out.append((name.value.decode(), mt.value, hd.value))
        return out

    def attn_store_enable(self, name, mode=1):
        self._chk(self.lib.rt_attn_store_enable(self.h, name.encode(), int(mode)))

    def attn_store_reset(self):
        self._chk(self.lib.rt_attn_store_reset(self.h))

    def attn_store_read(self, name):
        """-> (n_calls, map [1, rows, cols] on the GPU or None if nothing was recorded yet)"""
        import torch
        n, r, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.rt_attn_store_read(self.h, name.encode(), None, C.byref(n), C.byref(r), C.byref(c)))
        if r.value == 0:
            return n.value, None
        out = torch.empty(1, r.value, c.value, device=f"cuda:{self.device}")
        self._chk(self.lib.rt_attn_store_read(self.h, name.encode(), _ptr(out), C.byref(n), C.byref(r), C.byref(c)))
        return n.value, out

    def state_ptrs(self):
        """Device pointers of the sampler state: (latents [4,h,w], CFG-combined noise_pred of the last rich step)."""
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.rt_get_state_ptrs(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- profiling (HIP events around every MFMA kernel launch on the engine stream)
    PROF_CLASSES = {0: "gemm_kernel<A_DENSE>", 1: "gemm_kernel<A_CONV3*>", 2: "attn_kernel<self>", 3: "attn_kernel<cross>",
                    5: "gemm16_kernel<EPI_XATTN> (to_q + cross-attention)", 6: "xblock_kernel (to_q + cross-attention + to_out)"}
    PROF_STORE = 4          # attn_store_kernel (plain pass, token-map capture): priced in algorithmic HBM bytes

    def profile_read_store(self):
        n, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.rt_profile_read2(self.h, self.PROF_STORE, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return dict(launches=n.value, total_ms=ms.value, total_flops=fl.value, total_bytes=by.value)

    def profile_enable(self, on=True):
        self._chk(self.lib.rt_profile_enable(self.h, int(on)))

    def profile_read(self):
        out = {}
        for cls, name in self.PROF_CLASSES.items():
            n, ms, fl = C.c_int(), C.c_double(), C.c_double()
            self._chk(self.lib.rt_profile_read(self.h, cls, C.byref(n), C.byref(ms), C.byref(fl)))
            out[name] = dict(launches=n.value, total_ms=ms.value, total_flops=fl.value)
        return out

    def read_latents(self, h, w, with_ref=False):
        import torch
        out = torch.empty(1, 4, h, w, device=f"cuda:{self.device}")
        ref = torch.empty_like(out) if with_ref else None
        self._chk(self.lib.rt_get_latents(self.h, _ptr(out), _ptr(ref)))
        self.synchronize()
        return (out, ref) if with_ref else out

    # ---- hot path
    def region_step(self, i, guidance_scale, inject_selfattn=0.0, inject_background=0.0, xl=True, elide=False, defer_blend=False):
        self._chk(self.lib.rt_region_step(self.h, i, C.c_float(guidance_scale), C.c_double(inject_selfattn),
                                          C.c_double(inject_background), int(xl), int(bool(elide)) | (2 if defer_blend else 0)))

    # ---- intra-image split of a step over the ranks of a process group (launcher.split_region_step drives these)
    def region_step_part(self, i, guidance_scale, inject_selfattn, inject_background, xl, part, nparts, elide=False, defer_blend=False):
        """The UNet forwards of this rank's contiguous stream range of step i -> (first_stream, n_streams, (streams of the step, its
        text_ref stream or -1, injection on))."""
        first, count, info = C.c_int(), C.c_int(), (C.c_int * 3)()
        self._chk(self.lib.rt_region_step_part(self.h, i, C.c_float(guidance_scale), C.c_double(inject_selfattn), C.c_double(inject_background),
                                               int(xl), int(bool(elide)) | (2 if defer_blend else 0), part, nparts, C.byref(first), C.byref(count), info))
        return first.value, count.value, (info[0], info[1], bool(info[2]))

    def region_step_finish(self, i, guidance_scale, inject_selfattn, inject_background, xl, elide=False, defer_blend=False):
        self._chk(self.lib.rt_region_step_finish(self.h, i, C.c_float(guidance_scale), C.c_double(inject_selfattn), C.c_double(inject_background),
                                                 int(xl), int(bool(elide)) | (2 if defer_blend else 0)))

    def eps_info(self):
        """(device pointer, bytes per stream, max streams) of the noise-prediction buffer [max_streams][h*w][4] fp32."""
        p, b, m = C.c_void_p(), C.c_uint64(), C.c_int()
        self._chk(self.lib.rt_eps_info(self.h, C.byref(p), C.byref(b), C.byref(m)))
        return p.value, b.value, m.value

    def background_blend(self):
        self._chk(self.lib.rt_background_blend(self.h))

    def plain_step(self, i, guidance_scale):
        self._chk(self.lib.rt_plain_step(self.h, i, C.c_float(guidance_scale)))

    def plain_step_part(self, i, part, nparts):
        """The forwards of this rank's range of the plain step's streams [uncond, text] (launcher.split_plain_step); returns (first, count)."""
        first, count = C.c_int(0), C.c_int(0)
        self._chk(self.lib.rt_plain_step_part(self.h, i, part, nparts, C.byref(first), C.byref(count)))
        return first.value, count.value

    def plain_step_finish(self, i, guidance_scale):
        self._chk(self.lib.rt_plain_step_finish(self.h, i, C.c_float(guidance_scale)))

    def unet_forward(self, x, timestep, prompt_idx, in_scale=None, fontsize=None, qk_src=None, res_src=None):
        import torch
        x = x.contiguous().float()
        B, _, h, w = x.shape
        out = torch.empty_like(x)

        def iarr(v):
            return (C.c_int * B)(*[int(k) for k in v]) if v is not None else None
        sc = (C.c_float * B)(*[float(k) for k in in_scale]) if in_scale is not None else None
        self._chk(self.lib.rt_unet_forward(self.h, _ptr(x), B, h, w, C.c_float(float(timestep)), sc, iarr(prompt_idx),
                                           iarr(fontsize), iarr(qk_src), iarr(res_src), _ptr(out)))
        self.synchronize()
        return out


# ------------------------------------------------------------------------------------------------ VAE decoder
SD_VAE_CONFIG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
SDXL_VAE_CONFIG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32, scaling_factor=0.13025)


class RtVaeConfig(C.Structure):
    _fields_ = [("n_blocks", C.c_int), ("block_out_channels", C.c_int * RT_MAX_LEVELS), ("layers_per_block", C.c_int),
                ("norm_groups", C.c_int), ("scaling_factor", C.c_float), ("latent_h", C.c_int), ("latent_w", C.c_int),
                ("precise", C.c_int)]


class VaeDecoder:
    """AutoencoderKL decoder on the engine: `.decode(z)` (same call surface as diffusers' `vae.decode(z).sample`) and
    the colour-guidance update of the rich-text loop (rd.py:151-168 / xl.py:849-867).

    `precise=True` runs every contraction as three bf16 MFMA passes over (hi, lo) operand pairs - fp32-class products - which is
    what the SDXL pipeline of the reference asks of its VAE (xl.py:856 `.to(dtype=torch.float32)`); the SD pipeline decodes in the
    checkpoint's dtype (rd.py:160) and keeps the single-pass default."""

    def __init__(self, cfg, latent_h, latent_w, device=0, state_dict=None, precise=False):
        self.lib = load_library()
        c = RtVaeConfig()
        boc = tuple(cfg["block_out_channels"])
        c.n_blocks = len(boc)
        for i, v in enumerate(boc):
            c.block_out_channels[i] = v
        c.layers_per_block = cfg["layers_per_block"]
        c.norm_groups = cfg["norm_num_groups"]
        c.scaling_factor = cfg["scaling_factor"]
        c.latent_h, c.latent_w = latent_h, latent_w
        c.precise = int(bool(precise))
        self.precise = bool(precise)
        self.cfg, self.cfg_dict, self.device = c, dict(cfg), device
        self.scaling_factor = cfg["scaling_factor"]
        self.h = C.c_void_p()
        rc = self.lib.rt_vae_create(C.byref(c), device, C.byref(self.h))
        if rc != 0:
            raise RtError(rc, self.lib.rt_vae_last_error(None).decode())
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_vae_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_vae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def weight_table(self):
        n = self.lib.rt_vae_weight_count(self.h)
        name = C.create_string_buffer(256)
        shape, nd = (C.c_int64 * 4)(), C.c_int()
        out = []
        for i in range(n):
            self._chk(self.lib.rt_vae_weight_info(self.h, i, name, 256, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[k] for k in range(nd.value))))
        return out

    def load_state_dict(self, sd):
        import torch
        dev = f"cuda:{self.device}"
        for name, _ in self.weight_table():
            t = sd[name].to(dev).contiguous()
            dt = {torch.float32: DTYPE_F32, torch.float16: DTYPE_F16, torch.bfloat16: DTYPE_BF16}[t.dtype]
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            self._chk(self.lib.rt_vae_bind_weight(self.h, name.encode(), _ptr(t), dt, shape, t.dim()))
        self._chk(self.lib.rt_vae_synchronize(self.h))

    def arena(self):
        p, b = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.rt_vae_arena_info(self.h, C.byref(p), C.byref(b)))
        return p.value, b.value

    def arena_mark_bound(self):
        self._chk(self.lib.rt_vae_arena_mark_bound(self.h))

    def synchronize(self):
        self._chk(self.lib.rt_vae_synchronize(self.h))

    def decode(self, z, divide_by_scaling=False):
        """z [1,4,h,w] -> image [1,3,8h,8w] in [-1,1]"""
        import torch
        z = z.contiguous().float()
        _, _, h, w = z.shape
        out = torch.empty(1, 3, 8 * h, 8 * w, device=z.device)
        self._chk(self.lib.rt_vae_decode(self.h, _ptr(z), h, w, int(divide_by_scaling), _ptr(out)))
        return out

    def color_guidance(self, latents_ptr_or_tensor, noise_pred, alpha_t, h, w, color_obj_atten, target_rgb, weight, color_obj_atten_all,
                       want_grad=False):
        """In-place update of `latents` (tensor [1,4,h,w] or raw device pointer) exactly as rd.py:151-168."""
        import torch
        # the masks / targets are the same objects on every step of a loop (sample.py:87-88 builds them once): stage them on the
        # device once instead of per step (image-resolution masks from pageable host memory are a synchronous multi-MB copy)
        # the reference pairs masks and targets with zip() (rd.py:158): sample.py hands over n_color + 1 masks (get_token_maps
        # appends the background mask) but n_color targets, so the surplus mask is silently dropped.  Same here.
        n = min(len(color_obj_atten), len(target_rgb))
        if n == 0:
            raise RtError(-1, "color_guidance: no (mask, target RGB) pair")
        key = (id(color_obj_atten), id(target_rgb), id(color_obj_atten_all), n)
        if getattr(self, "_cg_key", None) != key:
            masks = torch.cat([m[:, 0].reshape(1, -1) for m in color_obj_atten[:n]]).contiguous().float().to(f"cuda:{self.device}")
            tgt = [float(v) for t in target_rgb[:n] for v in t.flatten().tolist()]
            if len(tgt) != 3 * n or masks.shape != (n, 64 * h * w):
                raise RtError(-1, f"color_guidance: need {n} RGB triples and [{n}, {64 * h * w}] masks, got {len(tgt)} values / {tuple(masks.shape)}")
            mall = color_obj_atten_all.contiguous().float().to(f"cuda:{self.device}")
            self._cg_key, self._cg_val = key, (masks, (C.c_float * len(tgt))(*tgt), mall, (color_obj_atten, target_rgb, color_obj_atten_all))
        masks, tgt_arr, mall, _ = self._cg_val
        grad = torch.empty(1, 4, h, w, device=f"cuda:{self.device}") if want_grad else None
        loss = C.c_float()
        lp = latents_ptr_or_tensor if isinstance(latents_ptr_or_tensor, int) else latents_ptr_or_tensor.data_ptr()
        npp = noise_pred if isinstance(noise_pred, int) else noise_pred.data_ptr()
        self._chk(self.lib.rt_vae_color_guidance(self.h, C.c_void_p(lp), C.c_void_p(npp), C.c_float(float(alpha_t)), h, w, _ptr(masks),
                                                 tgt_arr, n, C.c_float(float(weight)), _ptr(mall),
                                                 _ptr(grad), C.byref(loss)))
        return loss.value, grad
