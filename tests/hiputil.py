"""Thin torch<->C-ABI helpers for the GPU parity tests (call the product kernels through include/rtdiff.h)."""
import ctypes as C

import torch

from rich_text_to_image_amd.engine import load_library, _ptr

DEV = "cuda:0"


def chk(rc):
    if rc != 0:
        raise RuntimeError(f"rt_op error {rc}: {load_library().rt_op_last_error().decode()}")


def bf(t):
    return t.to(DEV).to(torch.bfloat16).contiguous()


def gemm(A, W, bias=None, epi=0, res=None, temb=None, rows_per_batch=0, mode=0, conv=None, out_cols=None):
    """A: bf16 [M,K] (dense) or NHWC [B,Hin,Win,Cin] (conv); W bf16 [N,K]."""
    lib = load_library()
    N, K = W.shape
    if mode == 0:
        M = A.shape[0]
        Hin = Win = Cin = Hout = Wout = 0
        lda = A.stride(0)
    else:
        B, Hin, Win, Cin = A.shape
        Hout, Wout = conv
        M = B * Hout * Wout
        rows_per_batch = Hout * Wout
        lda = 0
    oc = out_cols if out_cols is not None else (N // 2 if epi == 3 else N)
    out = torch.empty(M, oc, device=DEV, dtype={1: torch.float32, 4: torch.float16}.get(epi, torch.bfloat16))
    chk(lib.rt_op_gemm(_ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(res), _ptr(temb), mode, epi, M, N, K, lda, W.stride(0),
                       out.stride(0), res.stride(0) if res is not None else 0, temb.stride(0) if temb is not None else 0,
                       rows_per_batch, Hin, Win, Cin, Hout, Wout, None))
    torch.cuda.synchronize()
    return out


def attention(Q, K, VT, B, H, N, NK, DP, ldq=None, ldk=None, q_src=None, k_src=None, v_src=None, cross=False, wabs=None,
              wsgn=None, wset=None, nk_valid=None):
    lib = load_library()
    O = torch.zeros(B * N, H * DP, device=DEV, dtype=torch.bfloat16)

    def ia(v):
        return (C.c_int * B)(*v) if v is not None else None
    chk(lib.rt_op_attention(_ptr(Q), ldq or Q.stride(0), _ptr(K), ldk or K.stride(0), _ptr(VT), VT.stride(0), _ptr(O), O.stride(0),
                            ia(q_src), ia(k_src), ia(v_src), ia(wset), _ptr(wabs), _ptr(wsgn), B, H, N, NK,
                            nk_valid if nk_valid is not None else NK, DP, int(cross), None))
    torch.cuda.synchronize()
    return O


def groupnorm(x1, x2, G, gamma, beta, eps, silu, want_raw=False):
    lib = load_library()
    in_bf16 = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[x1.dtype]
    B, HW, C1 = x1.shape
    C2 = x2.shape[2] if x2 is not None else 0
    out = torch.empty(B, HW, C1 + C2, device=DEV, dtype=torch.bfloat16)
    raw = torch.empty_like(out) if want_raw else None
    chk(lib.rt_op_groupnorm(_ptr(x1), _ptr(x2), int(in_bf16), C1, C2, G, B, HW, _ptr(gamma), _ptr(beta), C.c_float(eps),
                            int(silu), _ptr(out), _ptr(raw), None))
    torch.cuda.synchronize()
    return (out, raw) if want_raw else out


def layernorm(x, gamma, beta, eps=1e-5):
    lib = load_library()
    rows, Cc = x.shape
    out = torch.empty(rows, Cc, device=DEV, dtype=torch.bfloat16)
    fn = lib.rt_op_layernorm_f16 if x.dtype == torch.float16 else lib.rt_op_layernorm
    chk(fn(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), rows, Cc, C.c_float(eps), None))
    torch.cuda.synchronize()
    return out


def small_linear(a, W, bias, silu_in=False):
    lib = load_library()
    B, K = a.shape
    N = W.shape[0]
    out = torch.zeros(B, N, device=DEV)
    chk(lib.rt_op_small_linear(_ptr(a), a.stride(0), _ptr(W), W.stride(0), _ptr(bias), _ptr(out), N, B, N, K, int(silu_in), 0, None))
    torch.cuda.synchronize()
    return out


def timestep_embed(t, dim):
    lib = load_library()
    out = torch.zeros(t.numel(), dim, device=DEV)
    chk(lib.rt_op_timestep_embed(_ptr(t), t.numel(), dim, _ptr(out), dim, None))
    torch.cuda.synchronize()
    return out


def report(name, got, ref, atol, rtol):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    msg = (f"{name}: max|err|={err.max().item():.4e} at {tuple(int(v) for v in torch.nonzero(err == err.max())[0])} "
           f"ref_rms={ref.pow(2).mean().sqrt().item():.4e} rel_l2={(err.pow(2).sum() / ref.pow(2).sum().clamp_min(1e-30)).sqrt().item():.4e} "
           f"bad={int(bad.sum())}/{bad.numel()}")
    print(msg)
    assert not bad.any(), msg
