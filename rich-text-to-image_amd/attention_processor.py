"""Operator seam of SURVEY 8b: an `AttnProcessor` for the REFERENCE's own `Attention` modules
(models/attention_processor.py:476-545, installed with `unet.set_attn_processor(HipAttnProcessor())`,
unet_2d_condition.py:570-626) assembled from the C-ABI operators `rt_op_gemm` / `rt_op_attention`.

Call contract kept: `processor(attn, hidden_states, real_attn_probs=None, attn_weights=None, encoder_hidden_states=None,
attention_mask=None, temb=None) -> (hidden_states, [probs_avg, probs])`.
* element [1][1] (per-head probabilities, 5.87 GB at SDXL in the reference) is an `AttnMapHandle`: it carries the bf16
  Q/K the probabilities are a function of; handing it back as `real_attn_probs` (what the reference's injection hooks do,
  rd.py:331,366,382) runs attention with (Q_ref, K_ref, V_current), which equals `bmm(P_ref, V)`.
  It answers `.shape` / `.detach()` like the tensor the hooks expect (rd.py:326-331).
  A caller that passes a REAL probability tensor [B*H, N, K] instead (attention_processor.py:522-524 accepts any tensor) gets
  `bmm(P, V)` per (batch entry, head) through `rt_op_gemm` on the bf16-cast probabilities, and the tensor itself back as [1][1].
* element [1][0] (head-averaged probabilities, consumed only by the token-map hooks rd.py:414-426, xl.py:980-992) is computed
  lazily on first tensor use (`rt_op_attention_probs_avg`, any key count: 64x64 SDXL self-attention maps included).
No torch arithmetic on the activation path: casts, projections, softmax, PV and the residual are library calls; torch
allocates buffers and reshapes views.  Weights are packed once per module (head padding, softmax scale folded into to_q).
"""
import ctypes as C
import math

import torch

from .engine import RtError, load_library


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dp(d):
    for c in (32, 64, 96, 160):
        if d <= c:
            return c
    raise ValueError(f"head dim {d} > 160 is not supported by the attention kernel")


class AttnMapHandle:
    """Stands for the [B*H, N, NK] probability tensor without materialising it."""

    def __init__(self, Q, K, B, H, N, NK):
        self.Q, self.K, self.B, self.H, self.N, self.NK = Q, K, B, H, N, NK
        self.shape = torch.Size((B * H, N, NK))

    def detach(self):
        return self


class LazyProbsAvg:
    """[B, N, NK] head-averaged probabilities, computed by the attention-store kernel when first used as a tensor."""

    def __init__(self, proc, Q, K, B, H, N, NK, NKpad, DP, cross):
        self._args = (proc, Q, K, B, H, N, NK, NKpad, DP, cross)
        self.shape = torch.Size((B, N, NK))
        self._t = None

    def tensor(self):
        if self._t is None:
            proc, Q, K, B, H, N, NK, NKpad, DP, cross = self._args
            out = torch.empty(B, N, NK, device=Q.device, dtype=torch.float32)
            # K holds NKpad rows per batch entry (cross: 77 keys padded to 96; self: NKpad == NK rows).  The store kernel wants the
            # key count padded to a multiple of 32 and the number of key ROWS that exist: a self-attention map over a token grid
            # that is a multiple of 8 but not of 32 (20x20 = 400) pads the count, not the rows
            pad32 = (NKpad + 31) // 32 * 32
            for b in range(B):
                proc._chk(proc.lib.rt_op_attention_probs_avg(_ptr(Q), Q.stride(0), b * N, _ptr(K), K.stride(0), b * NKpad, _ptr(out[b]),
                                                             H, N, NK, pad32, NKpad, DP, 0, None))
            self._t = out
        return self._t

    def detach(self):
        return self.tensor()

    def __getattr__(self, name):             # .cpu(), indexing helpers ... forward to the real tensor
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]


class HeadMeanOfTensor:
    """probs_avg when the caller supplied the probabilities itself: `reshape_batch_dim_to_heads_and_average`
    (attention_processor.py:166-171) of that tensor, evaluated only if somebody reads it (nobody does on the injection path)."""

    def __init__(self, probs, B, H):
        self._p, self._B, self._H, self._t = probs, B, H, None
        self.shape = torch.Size((B, probs.shape[1], probs.shape[2]))

    def tensor(self):
        if self._t is None:
            p = self._p
            self._t = p.reshape(self._B, self._H, p.shape[1], p.shape[2]).float().mean(1)
        return self._t

    def detach(self):
        return self.tensor()

    def __getattr__(self, name):
        return getattr(self.tensor(), name)

    def __getitem__(self, idx):
        return self.tensor()[idx]


class HipAttnProcessor:
    def __init__(self):
        self.lib = load_library()
        self._packed = {}

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, self.lib.rt_op_last_error().decode())

    # ---- one-time weight packing per Attention module (the layouts of DESIGN.md section 3)
    def _pack(self, attn, dev):
        key = id(attn)
        if key in self._packed:
            return self._packed[key]
        H = attn.heads
        inner = attn.to_q.weight.shape[0]
        d = inner // H
        DP = _dp(d)

        def heads_out(w, scale=1.0):                       # [H*d, Cin] -> bf16 [H*DP, Cin]
            o = torch.zeros(H, DP, w.shape[1])
            o[:, :d] = w.detach().float().cpu().reshape(H, d, -1) * scale
            return o.reshape(H * DP, -1).to(torch.bfloat16).to(dev).contiguous()

        def heads_bias(b, scale=1.0):
            if b is None:
                return None
            o = torch.zeros(H, DP)
            o[:, :d] = b.detach().float().cpu().reshape(H, d) * scale
            return o.reshape(-1).to(dev).contiguous()
        qs = attn.scale * math.log2(math.e)                # exp2 softmax: fold d^-1/2 * log2(e) into to_q
        wo = torch.zeros(attn.to_out[0].weight.shape[0], H, DP)
        wo[:, :, :d] = attn.to_out[0].weight.detach().float().cpu().reshape(-1, H, d)
        pk = dict(H=H, d=d, DP=DP,
                  wq=heads_out(attn.to_q.weight, qs), bq=heads_bias(attn.to_q.bias, qs),
                  wk=heads_out(attn.to_k.weight), bk=heads_bias(attn.to_k.bias),
                  wv=heads_out(attn.to_v.weight), bv=heads_bias(attn.to_v.bias),
                  wo=wo.reshape(wo.shape[0], H * DP).to(torch.bfloat16).to(dev).contiguous(),
                  bo=None if attn.to_out[0].bias is None else attn.to_out[0].bias.detach().float().to(dev).contiguous())
        self._packed[key] = pk
        return pk

    def _gemm(self, A, W, bias, out, epi=0, res=None):
        M, K = A.shape
        N = W.shape[0]
        self._chk(self.lib.rt_op_gemm(_ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(res), None, 0, epi, M, N, K, A.stride(0), W.stride(0),
                                      out.stride(0), res.stride(0) if res is not None else 0, 0, 0, 0, 0, 0, 0, 0, None))

    def _bf16(self, x2d):
        if x2d.dtype == torch.bfloat16:
            return x2d.contiguous()
        x2d = x2d.float().contiguous()
        out = torch.empty(x2d.shape, device=x2d.device, dtype=torch.bfloat16)
        self._chk(self.lib.rt_op_cast_bf16(_ptr(x2d), _ptr(out), x2d.numel(), None))
        return out

    def __call__(self, attn, hidden_states, real_attn_probs=None, attn_weights=None, encoder_hidden_states=None,
                 attention_mask=None, temb=None):
        if attention_mask is not None or getattr(attn, "spatial_norm", None) is not None or getattr(attn, "group_norm", None) is not None \
                or (encoder_hidden_states is not None and getattr(attn, "norm_cross", False)):
            raise NotImplementedError("HipAttnProcessor covers the rich-text path: no attention mask, spatial/group norm or norm_cross")
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b_, c_, h_, w_ = hidden_states.shape
            hidden_states = hidden_states.view(b_, c_, h_ * w_).transpose(1, 2)
        B, N, Cc = hidden_states.shape
        dev = hidden_states.device
        pk = self._pack(attn, dev)
        H, DP = pk["H"], pk["DP"]
        HD = H * DP
        x = self._bf16(hidden_states.reshape(B * N, Cc))
        cross = encoder_hidden_states is not None
        if cross:
            NK, Dc = encoder_hidden_states.shape[1], encoder_hidden_states.shape[2]
            NKp = ((NK + 31) // 32) * 32                                  # 77 -> 96
            ctx = torch.zeros(B, NKp, Dc, device=dev, dtype=torch.float32)
            ctx[:, :NK].copy_(encoder_hidden_states)
            kv_in = self._bf16(ctx.reshape(B * NKp, Dc))
        else:
            if N % 8:
                raise NotImplementedError("self-attention needs a token count that is a multiple of 8")
            NK = NKp = N
            kv_in = x
        real_tensor = None
        if isinstance(real_attn_probs, AttnMapHandle):
            Q, K = real_attn_probs.Q, real_attn_probs.K                   # injection: attend with the captured Q/K
        elif real_attn_probs is not None:
            if not torch.is_tensor(real_attn_probs) or tuple(real_attn_probs.shape) != (B * H, N, NK):
                raise TypeError(f"real_attn_probs must be a [{B * H}, {N}, {NK}] probability tensor (attention_processor.py:522-524) "
                                "or the AttnMapHandle a previous call of this processor returned")
            real_tensor = real_attn_probs
            Q = K = None
        else:
            Q = torch.empty(B * N, HD, device=dev, dtype=torch.bfloat16)
            K = torch.empty(B * NKp, HD, device=dev, dtype=torch.bfloat16)
            self._gemm(x, pk["wq"], pk["bq"], Q)
            self._gemm(kv_in, pk["wk"], pk["bk"], K)
        VT = torch.empty(HD, B * NKp, device=dev, dtype=torch.bfloat16)   # V^T = W_v X^T straight from the GEMM
        if pk["bv"] is not None:
            raise NotImplementedError("to_v bias is not used by the reference UNets")
        self._gemm(pk["wv"], kv_in, None, VT)
        wabs = wsgn = None
        wset = [0] * B
        if cross and attn_weights is not None:
            assert NK == 77                                               # attention_processor.py:386
            wabs = torch.zeros(2, NKp, device=dev); wabs[:, :NK] = 1.0
            wsgn = torch.ones(2, NKp, device=dev)
            fs = attn_weights['font_size'].to(dev).float()
            wabs[1, attn_weights['word_pos']] = fs.abs(); wsgn[1, attn_weights['word_pos']] = fs.sign()
            wset = [1] * B
        elif cross:
            wabs = torch.zeros(2, NKp, device=dev); wabs[:, :NK] = 1.0
            wsgn = torch.ones(2, NKp, device=dev)
        O = torch.empty(B * N, HD, device=dev, dtype=torch.bfloat16)
        ia = lambda v: (C.c_int * B)(*v)
        idx = list(range(B))
        if real_tensor is not None:
            # hidden = bmm(P, V) with the caller's probabilities (attention_processor.py:522-527): one MFMA GEMM per (b, h),
            # A = P[b*H+h] (bf16, keys zero-padded to NKp), W = the head's rows of V^T restricted to batch entry b's keys
            pf = torch.zeros(B * H, N, NKp, device=dev, dtype=torch.float32)
            pf[:, :, :NK].copy_(real_tensor)
            pb = self._bf16(pf.reshape(B * H * N, NKp))
            eb = 2                                                     # bytes per bf16
            for b in range(B):
                for h in range(H):
                    a_ptr = C.c_void_p(pb.data_ptr() + (b * H + h) * N * NKp * eb)
                    w_ptr = C.c_void_p(VT.data_ptr() + (h * DP * VT.stride(0) + b * NKp) * eb)
                    o_ptr = C.c_void_p(O.data_ptr() + (b * N * HD + h * DP) * eb)
                    self._chk(self.lib.rt_op_gemm(a_ptr, w_ptr, None, o_ptr, None, None, 0, 0, N, DP, NKp, NKp, VT.stride(0), HD,
                                                  0, 0, 0, 0, 0, 0, 0, 0, None))
        else:
            self._chk(self.lib.rt_op_attention(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(VT), VT.stride(0), _ptr(O), O.stride(0),
                                               ia(idx), ia(idx), ia(idx), ia(wset), _ptr(wabs), _ptr(wsgn), B, H, N, NKp, NK, DP, int(cross), None))
        out = torch.empty(B * N, Cc, device=dev, dtype=torch.float32)
        res = None
        if getattr(attn, "residual_connection", False) and input_ndim != 4:
            res = residual.reshape(B * N, Cc).float().contiguous()
        self._gemm(O, pk["wo"], pk["bo"], out, epi=1, res=res)            # to_out[0] (+ residual); to_out[1] is Dropout(0)
        hs = out.reshape(B, N, Cc)
        if input_ndim == 4:
            hs = hs.transpose(-1, -2).reshape(b_, c_, h_, w_)
            if getattr(attn, "residual_connection", False):
                hs = hs + residual
        rof = getattr(attn, "rescale_output_factor", 1.0)
        if rof != 1.0:
            hs = hs / rof
        hs = hs.to(residual.dtype)
        if real_tensor is not None:
            return hs, [HeadMeanOfTensor(real_tensor, B, H), real_tensor]
        if cross and attn_weights is not None:
            avg = None                                                    # font-size maps are never captured by the reference hooks
        else:
            avg = LazyProbsAvg(self, Q, K, B, H, N, NK, NKp, DP, cross)
        return hs, [avg, AttnMapHandle(Q, K, B, H, N, NK)]
