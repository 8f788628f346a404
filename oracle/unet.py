"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the reference UNet.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
It restates, in plain functional torch fp32 driven by a `state_dict` with the reference's key
names, exactly the sub-graph of /root/reference/models that the SD-v1.5 and SDXL pipelines
execute:

  * UNet2DConditionModel.forward            models/unet_2d_condition.py:703-983
  * DownBlock2D / CrossAttnDownBlock2D /
    UNetMidBlock2DCrossAttn / CrossAttnUpBlock2D / UpBlock2D
                                            models/unet_2d_blocks.py:1019,867,518,2010,2159
  * ResnetBlock2D.forward (+inject_states)  models/resnet.py:591-645
  * Upsample2D / Downsample2D               models/resnet.py:137-172, 213-222
  * Transformer2DModel.forward              models/transformer_2d.py:270-310
  * BasicTransformerBlock.forward, GEGLU FF models/attention.py:131-206, 209-304
  * Attention + AttnProcessor.__call__ and the font-size softmax
                                            models/attention_processor.py:326-407, 476-545
  * Timesteps / TimestepEmbedding           diffusers 0.18.2 models/embeddings.py ([memory],
                                            third-party, not on disk: parity unpinned)

Pinned against the reference itself (imported through oracle/refshim in the build container):
tests/test_oracle_vs_reference.py and the fixtures written by oracle/make_golden.py.

Hook semantics (models/region_diffusion.py:313-395,465-494; region_diffusion_sdxl.py:1018-1140)
are expressed through the `ctl` argument of `forward`:
  ctl = {"fontsize": {"word_pos": LongTensor, "font_size": FloatTensor} | None,   # attn2 only
         "capture": dict | None,   # filled with per-head attn1 probs + 'up_blocks.1.resnets.1' feature
         "inject":  dict | None}   # same keys: attn1 probs replace softmax, resnet feature injected
"""
import math

import torch
import torch.nn.functional as F

SD15_CONFIG = dict(
    in_channels=4, out_channels=4,
    block_out_channels=(320, 640, 1280, 1280),
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    layers_per_block=2, transformer_layers_per_block=(1, 1, 1, 1), attention_head_dim=(8, 8, 8, 8),
    cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5, use_linear_projection=False,
    addition_embed_type=None, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None,
)

SDXL_CONFIG = dict(
    in_channels=4, out_channels=4,
    block_out_channels=(320, 640, 1280),
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    layers_per_block=2, transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20),
    cross_attention_dim=2048, norm_num_groups=32, norm_eps=1e-5, use_linear_projection=True,
    addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
)

# Small configs with the same topology (3-level XL-like / 4-level SD-like) for fast traces.
# Channel counts are multiples of 32 and head dims of 8 so every engine kernel path is exercised.
TINY_XL_CONFIG = dict(
    in_channels=4, out_channels=4,
    block_out_channels=(32, 64, 128),
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    layers_per_block=2, transformer_layers_per_block=(1, 1, 2), attention_head_dim=(1, 2, 2),
    cross_attention_dim=64, norm_num_groups=8, norm_eps=1e-5, use_linear_projection=True,
    addition_embed_type="text_time", addition_time_embed_dim=8, projection_class_embeddings_input_dim=32 + 6 * 8,
)

TINY_SD_CONFIG = dict(
    in_channels=4, out_channels=4,
    block_out_channels=(32, 64, 128, 128),
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    layers_per_block=2, transformer_layers_per_block=(1, 1, 1, 1), attention_head_dim=(4, 4, 4, 4),
    cross_attention_dim=48, norm_num_groups=8, norm_eps=1e-5, use_linear_projection=False,
    addition_embed_type=None, addition_time_embed_dim=None, projection_class_embeddings_input_dim=None,
)

INJECT_RESNET = "up_blocks.1.resnets.1"     # region_diffusion.py:350, region_diffusion_sdxl.py:1101


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0):
    """diffusers 0.18.2 get_timestep_embedding ([memory]); UNet uses flip_sin_to_cos=True, shift 0
    (unet_2d_condition.py:284,379)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def attention_probs(q, k, scale, fontsize=None):
    """get_attention_scores, attention_processor.py:359-407. q,k: [B*H, N, d]."""
    scores = scale * torch.bmm(q, k.transpose(-1, -2))
    if fontsize is not None:
        assert k.shape[1] == 77                                            # :388
        stable = scores - scores.max(-1, True)[0]                           # :389
        e = stable.float().exp()                                            # :390
        fs_abs, fs_sign = fontsize["font_size"].abs(), fontsize["font_size"].sign()
        e[:, :, fontsize["word_pos"]] = e[:, :, fontsize["word_pos"]].clone() * fs_abs   # :393
        p = e / e.sum(-1, True)                                             # :395
        p[:, :, fontsize["word_pos"]] *= fs_sign                            # :396
        return p
    return scores.softmax(dim=-1)                                           # :401


class OracleUNet:
    def __init__(self, config, state_dict):
        self.cfg = dict(config)
        self.sd = {k: v.detach().float() for k, v in state_dict.items()}
        c = self.cfg
        nlev = len(c["block_out_channels"])
        for key in ("transformer_layers_per_block", "attention_head_dim"):
            if isinstance(c[key], int):
                c[key] = (c[key],) * nlev
        if isinstance(c["layers_per_block"], int):
            c["layers_per_block"] = (c["layers_per_block"],) * nlev

    # ---- leaf ops -------------------------------------------------------------------------
    def _lin(self, x, name, bias=True):
        return F.linear(x, self.sd[name + ".weight"], self.sd.get(name + ".bias") if bias else None)

    def _conv(self, x, name, stride=1, padding=1):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding)

    def _gn(self, x, name, eps):
        return F.group_norm(x, self.cfg["norm_num_groups"], self.sd[name + ".weight"], self.sd[name + ".bias"], eps)

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5)

    # ---- attention (attention_processor.py:476-545) -----------------------------------------
    # `attn_impl(oracle, name, x, heads, ctx, fontsize, real_probs) -> (out, probs)`: optional replacement of the whole attention
    # module (tests plug the product's AttnProcessor into this UNet the way set_attn_processor does in the reference)
    attn_impl = None

    def _attention(self, name, x, heads, ctx=None, fontsize=None, real_probs=None, capture=None):
        if self.attn_impl is not None:
            o, probs = self.attn_impl(self, name, x, heads, ctx, fontsize, real_probs)
            if capture is not None:
                capture[name] = probs
            return o, probs
        B, N, C = x.shape
        q = self._lin(x, name + ".to_q", bias=False)
        src = x if ctx is None else ctx
        k = self._lin(src, name + ".to_k", bias=False)
        v = self._lin(src, name + ".to_v", bias=False)
        d = C // heads

        def h2b(t):  # head_to_batch_dim :347-356
            return t.reshape(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)
        q, k, v = h2b(q), h2b(k), h2b(v)
        if real_probs is None:
            probs = attention_probs(q, k, d ** -0.5, fontsize)
        else:
            probs = real_probs                                               # :522-524
        o = torch.bmm(probs, v)
        o = o.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)   # batch_to_head_dim
        o = self._lin(o, name + ".to_out.0")
        if capture is not None:
            capture[name] = probs.detach()
        return o, probs

    # ---- transformer (transformer_2d.py:270-310, attention.py:131-206) ----------------------
    def _transformer(self, name, x, ctx, heads, nlayers, ctl, store=None):
        B, C, H, W = x.shape
        res = x
        h = self._gn(x, name + ".norm", 1e-6)                                # transformer_2d.py:137
        if not self.cfg["use_linear_projection"]:
            h = self._conv(h, name + ".proj_in", padding=0)
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        else:
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = self._lin(h, name + ".proj_in")
        for li in range(nlayers):
            bn = f"{name}.transformer_blocks.{li}"
            a1 = bn + ".attn1"
            inj = ctl.get("inject") if ctl else None
            cap = ctl.get("capture") if ctl else None
            real = inj[a1] if (inj is not None and a1 in inj) else None
            n1 = self._ln(h, bn + ".norm1")
            o, p1 = self._attention(a1, n1, heads, real_probs=real, capture=cap)
            if store is not None:
                store(a1, p1, heads)
            h = o + h
            n2 = self._ln(h, bn + ".norm2")
            fs = ctl.get("fontsize") if ctl else None
            o, p2 = self._attention(bn + ".attn2", n2, heads, ctx=ctx, fontsize=fs)
            if store is not None:
                store(bn + ".attn2", p2, heads)
            h = o + h
            n3 = self._ln(h, bn + ".norm3")
            g = self._lin(n3, bn + ".ff.net.0.proj")
            a, gate = g.chunk(2, dim=-1)                                     # attention.py:300-304
            g = a * F.gelu(gate)
            h = self._lin(g, bn + ".ff.net.2") + h
        if not self.cfg["use_linear_projection"]:
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
            h = self._conv(h, name + ".proj_out", padding=0)
        else:
            h = self._lin(h, name + ".proj_out")
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        return h + res

    # ---- resnet (resnet.py:591-645) -----------------------------------------------------------
    def _resnet(self, name, x, emb, ctl):
        eps = self.cfg["norm_eps"]
        h = F.silu(self._gn(x, name + ".norm1", eps))
        h = self._conv(h, name + ".conv1")
        t = self._lin(F.silu(emb), name + ".time_emb_proj")[:, :, None, None]
        h = h + t
        h = F.silu(self._gn(h, name + ".norm2", eps))
        h = self._conv(h, name + ".conv2")
        if (name + ".conv_shortcut.weight") in self.sd:
            x = self._conv(x, name + ".conv_shortcut", padding=0)
        inj = ctl.get("inject") if ctl else None
        cap = ctl.get("capture") if ctl else None
        if cap is not None and name == INJECT_RESNET:
            cap[name] = h.detach()
        if inj is not None and name in inj:
            return x + inj[name]                                             # :639-641
        return x + h

    # ---- forward (unet_2d_condition.py:703-983) -----------------------------------------------
    def forward(self, sample, timestep, ctx, added=None, ctl=None, store=None):
        """sample [B,4,h,w]; timestep scalar; ctx [B,77,D]; added = {"text_embeds","time_ids"}.
        `store(name, per_head_probs[B*H,N,K], heads)` is called for every attention module (token-map hooks)."""
        c = self.cfg
        B = sample.shape[0]
        boc = c["block_out_channels"]
        t = torch.as_tensor(timestep).reshape(-1).expand(B)
        emb = timestep_embedding(t, boc[0])
        emb = self._lin(F.silu(self._lin(emb, "time_embedding.linear_1")), "time_embedding.linear_2")
        if c["addition_embed_type"] == "text_time":                          # :841-857
            tid = timestep_embedding(added["time_ids"].flatten(), c["addition_time_embed_dim"])
            tid = tid.reshape(added["text_embeds"].shape[0], -1)
            add = torch.cat([added["text_embeds"].float(), tid], dim=-1)
            aug = self._lin(F.silu(self._lin(add, "add_embedding.linear_1")), "add_embedding.linear_2")
            emb = emb + aug
        x = self._conv(sample.float(), "conv_in")
        skips = [x]
        nlev = len(boc)
        for i, bt in enumerate(c["down_block_types"]):
            heads = c["attention_head_dim"][i]
            for j in range(c["layers_per_block"][i]):
                x = self._resnet(f"down_blocks.{i}.resnets.{j}", x, emb, ctl)
                if bt == "CrossAttnDownBlock2D":
                    x = self._transformer(f"down_blocks.{i}.attentions.{j}", x, ctx, heads,
                                          c["transformer_layers_per_block"][i], ctl, store)
                skips.append(x)
            if i != nlev - 1:
                x = self._conv(x, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1)
                skips.append(x)
        heads = c["attention_head_dim"][-1]
        x = self._resnet("mid_block.resnets.0", x, emb, ctl)
        x = self._transformer("mid_block.attentions.0", x, ctx, heads, c["transformer_layers_per_block"][-1], ctl, store)
        x = self._resnet("mid_block.resnets.1", x, emb, ctl)
        rev_heads = list(reversed(c["attention_head_dim"]))
        rev_tl = list(reversed(c["transformer_layers_per_block"]))
        rev_lpb = list(reversed(c["layers_per_block"]))
        for i, bt in enumerate(c["up_block_types"]):
            for j in range(rev_lpb[i] + 1):
                x = torch.cat([x, skips.pop()], dim=1)
                x = self._resnet(f"up_blocks.{i}.resnets.{j}", x, emb, ctl)
                if bt == "CrossAttnUpBlock2D":
                    x = self._transformer(f"up_blocks.{i}.attentions.{j}", x, ctx, rev_heads[i], rev_tl[i], ctl, store)
            if i != nlev - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")       # resnet.py:160
                x = self._conv(x, f"up_blocks.{i}.upsamplers.0.conv")
        x = F.silu(self._gn(x, "conv_norm_out", c["norm_eps"]))
        return self._conv(x, "conv_out")


def reference_kwargs(config):
    """kwargs for the reference's UNet2DConditionModel(...) matching an oracle config."""
    kw = {k: v for k, v in config.items() if v is not None}
    return kw


def random_state_dict(config, seed=0, ref_unet_cls=None):
    """Random-init weights with the reference's key names and shapes.

    Without the reference class (GPU box) the shapes are derived from the config by
    `engine_weight_shapes`; values use the same uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) family as
    torch's default init so activations stay O(1)."""
    from .shapes import weight_shapes
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in weight_shapes(config).items():
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            bound = 1.0 / math.sqrt(fan_in)
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.endswith(".weight"):          # 1-D weight => norm scale
            sd[name] = 1.0 + 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
        else:
            sd[name] = 0.05 * (torch.rand(shape, generator=g) * 2 - 1)
    return sd
