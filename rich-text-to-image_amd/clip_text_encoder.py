"""CLIP text encoders on the engine's operators: replaces transformers' `CLIPTextModel` /
`CLIPTextModelWithProjection` as the reference calls them (models/region_diffusion.py:53-66: `text_encoder(ids)[0]`;
models/region_diffusion_sdxl.py:330-356: `out[0]` = pooled/projected embedding, `out.hidden_states[-2]`).

Every contraction is `rt_op_gemm` (bf16 MFMA, fp32 accumulate), LayerNorm is `rt_op_layernorm`, the causal 77-token attention,
the token/position embedding and the MLP activation are the small kernels of `csrc/text.hip`; the residual stream is fp32.
Weights come under the transformers state-dict names (`text_model.encoder.layers.N.self_attn.q_proj.weight`, ...).
torch allocates buffers, gathers the EOS rows and wraps results - no torch arithmetic on the path.
"""
import ctypes as C
import types

import torch

from .engine import RtError, load_library


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def empty_state_dict(config, with_projection=False, vocab_size=None):
    """Zero tensors under the transformers CLIPTextModel[WithProjection] key names with the shapes `config` implies: what a rank that
    receives the weights by broadcast constructs its encoder from (launcher.broadcast_pipeline needs equal shapes on every rank)."""
    g = (lambda k, d=None: config.get(k, d)) if isinstance(config, dict) else (lambda k, d=None: getattr(config, k, d))
    Cc, L, I, N = g("hidden_size"), g("num_hidden_layers"), g("intermediate_size"), g("max_position_embeddings", 77)
    V = vocab_size or g("vocab_size", 49408)
    z = lambda *s: torch.zeros(*s)
    sd = {"text_model.embeddings.token_embedding.weight": z(V, Cc), "text_model.embeddings.position_embedding.weight": z(N, Cc),
          "text_model.final_layer_norm.weight": z(Cc), "text_model.final_layer_norm.bias": z(Cc)}
    for i in range(L):
        q = f"text_model.encoder.layers.{i}."
        for n in "qkv":
            sd[q + f"self_attn.{n}_proj.weight"], sd[q + f"self_attn.{n}_proj.bias"] = z(Cc, Cc), z(Cc)
        sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"] = z(Cc, Cc), z(Cc)
        for n in ("layer_norm1", "layer_norm2"):
            sd[q + n + ".weight"], sd[q + n + ".bias"] = z(Cc), z(Cc)
        sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"] = z(I, Cc), z(I)
        sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"] = z(Cc, I), z(Cc)
    if with_projection:
        sd["text_projection.weight"] = z(g("projection_dim", Cc), Cc)
    return sd


class HipCLIPTextEncoder:
    def __init__(self, state_dict, config, device=0, with_projection=False):
        """config: transformers CLIPTextConfig or a dict with hidden_size, num_attention_heads, num_hidden_layers, intermediate_size,
        hidden_act, layer_norm_eps, eos_token_id, max_position_embeddings (, projection_dim)."""
        g = (lambda k, d=None: config.get(k, d)) if isinstance(config, dict) else (lambda k, d=None: getattr(config, k, d))
        self.C, self.H, self.L, self.I = g("hidden_size"), g("num_attention_heads"), g("num_hidden_layers"), g("intermediate_size")
        self.act = {"quick_gelu": 0, "gelu": 1}[g("hidden_act", "quick_gelu")]
        self.eps, self.eos, self.N = g("layer_norm_eps", 1e-5), g("eos_token_id", 2), g("max_position_embeddings", 77)
        self.with_projection = with_projection
        self.lib = load_library()
        self.dev = torch.device(f"cuda:{device}")
        if self.C % 8 or self.I % 8 or self.C > 1280:
            raise ValueError("hidden_size / intermediate_size must be multiples of 8 and hidden_size <= 1280")
        sd = {k: v.detach() for k, v in state_dict.items()}
        f32 = lambda k: sd[k].float().to(self.dev).contiguous()
        b16 = lambda t: t.to(torch.bfloat16).to(self.dev).contiguous()
        p = "text_model." if any(k.startswith("text_model.") for k in sd) else ""      # transformers 4.x checkpoints / 5.x CLIPTextModel
        self.tok, self.pos = f32(p + "embeddings.token_embedding.weight"), f32(p + "embeddings.position_embedding.weight")
        self.layers = []
        for i in range(self.L):
            q = f"{p}encoder.layers.{i}."
            wqkv = torch.cat([sd[q + f"self_attn.{n}_proj.weight"].float() for n in "qkv"])
            bqkv = torch.cat([sd[q + f"self_attn.{n}_proj.bias"].float() for n in "qkv"])
            self.layers.append(dict(
                ln1=(f32(q + "layer_norm1.weight"), f32(q + "layer_norm1.bias")), ln2=(f32(q + "layer_norm2.weight"), f32(q + "layer_norm2.bias")),
                wqkv=b16(wqkv), bqkv=bqkv.to(self.dev).contiguous(), wo=b16(sd[q + "self_attn.out_proj.weight"].float()), bo=f32(q + "self_attn.out_proj.bias"),
                w1=b16(sd[q + "mlp.fc1.weight"].float()), b1=f32(q + "mlp.fc1.bias"), w2=b16(sd[q + "mlp.fc2.weight"].float()), b2=f32(q + "mlp.fc2.bias")))
        self.lnf = (f32(p + "final_layer_norm.weight"), f32(p + "final_layer_norm.bias"))
        self.wproj = b16(sd["text_projection.weight"].float()) if with_projection else None
        self.dtype = torch.float32

    def parameter_tensors(self):
        """Every device-resident weight in a fixed order (launcher.broadcast_pipeline)."""
        out = [self.tok, self.pos]
        for ly in self.layers:
            out += [ly["ln1"][0], ly["ln1"][1], ly["ln2"][0], ly["ln2"][1], ly["wqkv"], ly["bqkv"], ly["wo"], ly["bo"], ly["w1"], ly["b1"],
                    ly["w2"], ly["b2"]]
        out += [self.lnf[0], self.lnf[1]]
        if self.wproj is not None:
            out.append(self.wproj)
        return out

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_op_last_error().decode())

    def _gemm(self, A, W, bias, out, epi=0, res=None):
        M, K = A.shape
        N = W.shape[0]
        self._chk(self.lib.rt_op_gemm(_ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(res), None, 0, epi, M, N, K, A.stride(0), W.stride(0),
                                      out.stride(0), res.stride(0) if res is not None else 0, 0, 0, 0, 0, 0, 0, 0, None))

    def _ln(self, x, wb):
        out = torch.empty(x.shape, device=self.dev, dtype=torch.bfloat16)
        self._chk(self.lib.rt_op_layernorm(_ptr(x), _ptr(wb[0]), _ptr(wb[1]), _ptr(out), x.shape[0], x.shape[1], C.c_float(self.eps), None))
        return out

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def __call__(self, input_ids, output_hidden_states=False, **_):
        ids = input_ids.to(self.dev).to(torch.int32).contiguous()
        B, N = ids.shape
        if N > 128:
            raise ValueError("at most 128 tokens")
        M, Cc, H = B * N, self.C, self.H
        d = Cc // H
        x = torch.empty(M, Cc, device=self.dev)
        self._chk(self.lib.rt_op_embed(_ptr(ids), _ptr(self.tok), _ptr(self.pos), _ptr(x), M, N, Cc, self.tok.shape[0], None))
        hidden = [x.reshape(B, N, Cc).clone()] if output_hidden_states else None
        qkv = torch.empty(M, 3 * Cc, device=self.dev, dtype=torch.bfloat16)
        att = torch.empty(M, Cc, device=self.dev, dtype=torch.bfloat16)
        f1 = torch.empty(M, self.I, device=self.dev, dtype=torch.bfloat16)
        f2 = torch.empty_like(f1)
        for ly in self.layers:
            h = self._ln(x, ly["ln1"])
            self._gemm(h, ly["wqkv"], ly["bqkv"], qkv)
            self._chk(self.lib.rt_op_causal_attention(_ptr(qkv), C.c_void_p(qkv.data_ptr() + 2 * Cc), C.c_void_p(qkv.data_ptr() + 4 * Cc), 3 * Cc,
                                                      _ptr(att), Cc, B, H, N, d, C.c_float(d ** -0.5), None))
            self._gemm(att, ly["wo"], ly["bo"], x, epi=1, res=x)
            h = self._ln(x, ly["ln2"])
            self._gemm(h, ly["w1"], ly["b1"], f1)
            self._chk(self.lib.rt_op_activation(_ptr(f1), _ptr(f2), f1.numel(), self.act, None))
            self._gemm(f2, ly["w2"], ly["b2"], x, epi=1, res=x)
            if output_hidden_states:
                hidden.append(x.reshape(B, N, Cc).clone())
        last = self._ln(x, self.lnf)                                            # bf16 [M, C]
        last_f = last.float().reshape(B, N, Cc)
        # pooled = final-LN features at the EOS token (transformers: argmax of the ids for the legacy eos id 2, else first eos position)
        ids64 = input_ids.to(self.dev).long()
        eos_pos = ids64.argmax(-1) if self.eos == 2 else (ids64 == self.eos).int().argmax(-1)
        rows = (torch.arange(B, device=self.dev) * N + eos_pos)
        pooled_b = last.index_select(0, rows).contiguous()                      # gather only
        out = types.SimpleNamespace(last_hidden_state=last_f, pooler_output=pooled_b.float(), hidden_states=tuple(hidden) if hidden else None)
        if self.with_projection:
            P = self.wproj.shape[0]
            pb = torch.zeros(max(B, 1), Cc, device=self.dev, dtype=torch.bfloat16); pb[:B] = pooled_b
            te = torch.empty(B, P, device=self.dev)
            self._gemm(pb[:B], self.wproj, None, te, epi=1)
            out.text_embeds = te
            first = te
        else:
            first = last_f
        torch.cuda.synchronize(self.dev)

        class _Out(tuple):
            pass
        res = _Out((first,))
        res.__dict__.update(out.__dict__)
        return res
