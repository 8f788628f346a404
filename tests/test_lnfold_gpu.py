"""LayerNorm folded into the projections that consume it (round 6; csrc/gemm16.hip "LNF"), kernel level, through the C ABI.

BasicTransformerBlock normalises the fp16 trunk in front of attn1 / attn2 / ff (/root/reference/models/attention.py:150,168,181:
`norm1`, `norm2`, `norm3`, nn.LayerNorm eps 1e-5 affine).  The engine no longer launches those LayerNorms at SDXL's widths: the trunk's
producer (fp16-trunk epilogue of proj_in / to_out / ff.net.2) leaves, next to the fp16 trunk, xb = the same values as bf16 and per-row
partial sums of xb; the consumer GEMM reads xb against W' = bf16(gamma W) and applies  rstd (xb W'^T - mu s) + c  in its epilogue
(GEGLU: in front of the gelu): LayerNorm of the bf16-rounded trunk, exactly.

Reference arithmetic: torch fp32 `F.layer_norm(x) @ W^T + b` on the SAME fp16 trunk and bf16 weights - what the reference module
computes - with the bf16-output tolerance of the other GEMM tests.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hiputil import DEV, bf, gemm, layernorm, report  # noqa: E402

BF16_OUT = dict(atol=2e-2, rtol=1.2e-2)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _lib():
    from rich_text_to_image_amd.engine import load_library
    return load_library()


def _ptr(t):
    from rich_text_to_image_amd.engine import _ptr as p
    return p(t)


def _trunk(tokens, Cw, seed, mean_scale=1.0):
    """An fp16 trunk with per-row means and scales that differ (what a residual stream looks like), incl. a few large-mean rows."""
    x = rnd(tokens, Cw, seed=seed) * (0.5 + rnd(tokens, 1, seed=seed + 1).abs() * 2) + rnd(tokens, 1, seed=seed + 2) * mean_scale
    x = x + rnd(1, Cw, seed=seed + 3) * 0.5                                   # channel-wise offsets (outlier channels)
    return x.to(DEV).to(torch.float16).contiguous()


def _partials_ref(xb, bn):
    """(sum, sum of squares) per row and column tile of bn columns, pair-major [tiles / 2][rows] float4 = two tiles."""
    rows = xb.shape[0]
    v = xb.double().reshape(rows, -1, bn)
    t = torch.stack([v.sum(-1), (v * v).sum(-1)], dim=-1)                     # [rows, tiles, 2]
    return t.reshape(rows, -1, 4).permute(1, 0, 2).contiguous()


def _emit(A, W, bias, res, rps):
    lib = _lib()
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    xb = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    part = torch.full((4, M, 4), float("nan"), device=DEV, dtype=torch.float32)
    bn = C.c_int(0)
    rc = lib.rt_op_gemm_emit_partials(_ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(res), M, N, K, rps, _ptr(xb), _ptr(part), C.byref(bn), None)
    assert rc == 0, lib.rt_op_last_error().decode()
    torch.cuda.synchronize()
    assert bn.value in (160, 320)
    return out, xb, part[:N // bn.value // 2].contiguous(), bn.value


def _standalone(x16, bn):
    lib = _lib()
    rows, Cw = x16.shape
    part = torch.empty(Cw // bn // 2, rows, 4, device=DEV, dtype=torch.float32)
    xb = torch.empty(rows, Cw, device=DEV, dtype=torch.bfloat16)
    assert lib.rt_op_ln_partials(_ptr(x16), _ptr(xb), _ptr(part), rows, Cw, bn, None) == 0, lib.rt_op_last_error().decode()
    torch.cuda.synchronize()
    return xb, part


def _ln_gemm(x16, gamma, beta, W, bias, epi=0, vt=False, rps=0, xb=None, part=None, bn=None):
    lib = _lib()
    tokens, Cw = x16.shape
    N = W.shape[0]
    if vt:
        out = torch.empty(N, tokens, device=DEV, dtype=torch.bfloat16)
    else:
        out = torch.empty(tokens, N // 2 if epi == 3 else N, device=DEV, dtype=torch.bfloat16)
    bn = bn or (160 if Cw == 1280 else 320)                                  # what the engine's producers of that width use at SDXL's sizes
    rc = lib.rt_op_ln_gemm(_ptr(x16), _ptr(gamma), _ptr(beta), _ptr(W), _ptr(bias), _ptr(out), tokens, N, Cw, epi, int(vt), rps, _ptr(xb), _ptr(part), bn, None)
    assert rc == 0, lib.rt_op_last_error().decode()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M,N,K,rps", [(7168, 1280, 1280, 1024), (7168, 1280, 5120, 1024), (28672, 640, 640, 4096), (2048, 1280, 1280, 1024),
                                       (8192, 640, 2560, 4096), (5000, 640, 640, 5000), (4096, 640, 640, 1024)])
def test_trunk_producer_leaves_layernorm_partials(M, N, K, rps):
    """The fp16-trunk epilogue with LNF = 2: the trunk it writes is the plain kernel's, xb is the bf16 rounding of the same fp32 values
    (within one bf16 ulp of the fp16 trunk, every element written), and the partials are the sums of xb over the column tiles of the producer's grid
    (float64 check to fp32 rounding).  The stand-alone kernel gives xb = bf16(fp16 trunk) and the sums of that.  7- and 2-stream batches of both
    SDXL levels (224x160 K-split, 224x320, 64x160, 128x320 / 64x320 tiles) and a ragged M."""
    A, W = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3).to(DEV)
    res = (rnd(M, N, seed=4) * 3 + rnd(M, 1, seed=5) * 2).to(DEV).to(torch.float16)
    out, xb, part, bn = _emit(A, W, bias, res, rps)
    plain = torch.empty_like(out)
    lib = _lib()
    assert lib.rt_op_gemm16_variant(_ptr(A), _ptr(W), _ptr(bias), _ptr(plain), _ptr(res), 4, M, N, K, K, K, N, N, 0, -1, 0, None) == 0
    torch.cuda.synchronize()
    # (variant -1 picks by the batch alone; the emitting launch keys on rows_per_stream: same CLASS => same bits)
    assert torch.equal(out, plain) or torch.allclose(out.float(), plain.float(), atol=4e-3, rtol=1.5e-3), "emitting epilogue changed the trunk"
    report(f"emit trunk {M}x{N}x{K}", out, A.float() @ W.float().t() + bias + res.float(), atol=4e-3, rtol=1.5e-3)
    assert not torch.isnan(part).any() and not torch.isnan(xb.float()).any(), "an element of xb / a (row, block) partial was not written"
    assert torch.allclose(xb.float(), out.float(), rtol=2.0 ** -7, atol=1e-4), "xb is not the bf16 rounding of the trunk's values"
    ref = _partials_ref(xb, bn)
    assert torch.allclose(part.double(), ref, rtol=2e-6, atol=2e-3), (part.double() - ref).abs().max().item()
    for bn2 in (160, 320):
        xb2, alone = _standalone(out, bn2)
        assert torch.equal(xb2, out.to(torch.bfloat16))
        ref2 = _partials_ref(xb2, bn2)
        assert torch.allclose(alone.double(), ref2, rtol=2e-6, atol=2e-3), (alone.double() - ref2).abs().max().item()
    out3, xb3, part3, _ = _emit(A, W, bias, res, rps)                        # deterministic
    assert torch.equal(out, out3) and torch.equal(part, part3) and torch.equal(xb, xb3)


def report_rows(name, got, ref, x16, atol, rtol, vt=False, geglu=None):
    """`report` with the tolerance of a token scaled by sqrt(1 + (mu / sigma)^2) of its trunk row: the fold multiplies bf16(x), not
    bf16(LN(x)), so the operand rounding of a row is relative to |x| instead of |x - mu| (csrc/gemm16.hip, "LNF")."""
    xf = x16.float()
    amp = (1 + (xf.mean(1) / xf.std(1, unbiased=False)).pow(2)).sqrt().cpu()
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    if geglu is None:
        tol = (atol + rtol * ref.abs()) * (amp[None, :] if vt else amp[:, None])
    else:                                                                    # ref = a gelu(g): both projections carry the row's tolerance, |gelu'| <= 1.13
        a, g = (t.float().cpu() for t in geglu)
        ta, tg = (atol + rtol * a.abs()) * amp[:, None], (atol + rtol * g.abs()) * amp[:, None]
        tol = ta * F.gelu(g).abs() + 1.13 * a.abs() * tg + ta * tg + 2.0 ** -8 * ref.abs() + 1e-3
    bad = err > tol
    print(f"{name}: max|err|={err.max().item():.4e} rel_l2={(err.pow(2).sum() / ref.pow(2).sum()).sqrt().item():.4e} "
          f"row amplification max {amp.max().item():.2f} median {amp.median().item():.2f} bad={int(bad.sum())}/{bad.numel()}")
    assert not bad.any()


def _geglu_rows(half):
    rows = []
    for blk in range(half // 32):
        rows += list(range(blk * 32, blk * 32 + 32)) + list(range(half + blk * 32, half + blk * 32 + 32))
    return rows


@pytest.mark.parametrize("tokens,Cw,rps", [(7168, 1280, 1024), (28672, 640, 4096), (2048, 1280, 1024), (8192, 640, 4096), (4096, 1280, 1024)])
def test_folded_consumers_against_layernorm_reference(tokens, Cw, rps):
    """to_q / stacked Q|K (bf16 out), V^T (tokens on the columns), GEGLU - each against fp32 `layer_norm(x) W^T + b` on the same fp16
    trunk / bf16 weights, and against the round-5 path (LayerNorm launch -> bf16 -> plain GEMM) for the size of the change."""
    x = _trunk(tokens, Cw, seed=11)
    gamma = (1.0 + 0.3 * rnd(Cw, seed=12)).to(DEV)
    beta = (0.2 * rnd(Cw, seed=13)).to(DEV)
    ln = F.layer_norm(x.float(), (Cw,), gamma, beta, 1e-5)
    n_old = layernorm(x, gamma, beta)                                        # what rounds 1 - 5 fed the projections
    # attn2.to_q (N = C) and attn1's stacked to_q | to_k (N = 2 C, bias-free in the reference; a bias exercises c = b + W beta)
    for N, seed in ((Cw, 21), (2 * Cw, 22)):
        W = bf(rnd(N, Cw, seed=seed, scale=Cw ** -0.5))
        b = rnd(N, seed=seed + 100).to(DEV)
        ref = ln @ W.float().t() + b
        got = _ln_gemm(x, gamma, beta, W, b, epi=0, rps=rps)
        report_rows(f"folded projection {tokens}x{N}x{Cw}", got, ref, x, **BF16_OUT)
        old = gemm(n_old, W, b, epi=0)
        e_new, e_old = (got.float() - ref).pow(2).mean().sqrt().item(), (old.float() - ref).pow(2).mean().sqrt().item()
        print(f"  rms error vs fp32: folded {e_new:.3e}, LayerNorm launch + bf16 GEMM {e_old:.3e}")
        assert e_new < 2.0 * e_old + 1e-4
        assert torch.equal(got, _ln_gemm(x, gamma, beta, W, b, epi=0, rps=rps))
    # attn1.to_v as V^T = Wv LN(x)^T (no bias in the reference: c = Wv beta)
    Wv = bf(rnd(Cw, Cw, seed=31, scale=Cw ** -0.5))
    got = _ln_gemm(x, gamma, beta, Wv, None, epi=0, vt=True, rps=rps)
    report_rows(f"folded V^T {Cw}x{tokens}x{Cw}", got, Wv.float() @ ln.t(), x, vt=True, **BF16_OUT)
    assert torch.equal(got, _ln_gemm(x, gamma, beta, Wv, None, epi=0, vt=True, rps=rps))
    # ff.net.0 (GEGLU): Linear(C, 8 C), a * gelu(g); rows interleaved per 64-block
    Wfull = rnd(8 * Cw, Cw, seed=41, scale=Cw ** -0.5)
    bfull = rnd(8 * Cw, seed=42)
    rows = _geglu_rows(4 * Cw)
    Wp, bp = bf(Wfull[rows]), bfull[rows].to(DEV).contiguous()
    h = ln @ bf(Wfull).float().t() + bfull.to(DEV)
    a, g = h.chunk(2, dim=-1)
    got = _ln_gemm(x, gamma, beta, Wp, bp, epi=3, rps=rps)
    report_rows(f"folded GEGLU {tokens}x{8 * Cw}x{Cw}", got, a * F.gelu(g), x, geglu=(a, g), **BF16_OUT)
    old = gemm(n_old, Wp, bp, epi=3)
    ref = a * F.gelu(g)
    print(f"  GEGLU rms error vs fp32: folded {(got.float() - ref).pow(2).mean().sqrt().item():.3e}, round-5 path {(old.float() - ref).pow(2).mean().sqrt().item():.3e}")
    assert torch.equal(got, _ln_gemm(x, gamma, beta, Wp, bp, epi=3, rps=rps))


def test_folded_consumer_takes_the_producers_partials_and_large_row_means():
    """Chain of the engine: to_out (leaves xb + partials) -> folded to_q, against fp32 layer_norm of the fp16 trunk.  Rows whose mean is
    30 standard deviations away are the hard case of the fold (the operand error sits on x, not on LN(x): sqrt(1 + mu^2 / sigma^2) times
    the bf16 rounding; var = E[x^2] - mu^2 in fp32): reported separately and held to a 30x looser tolerance than the rest."""
    M, Cw, rps = 7168, 1280, 1024
    A, Wo = bf(rnd(M, Cw, seed=1)), bf(rnd(Cw, Cw, seed=2, scale=Cw ** -0.5))
    bo = rnd(Cw, seed=3).to(DEV)
    hard = (torch.arange(M) % 7 == 0)
    res = (rnd(M, Cw, seed=4) + 30.0 * hard.float()[:, None]).to(DEV).to(torch.float16)
    trunk, xb, part, bn = _emit(A, Wo, bo, res, rps)
    gamma, beta = (1.0 + 0.3 * rnd(Cw, seed=12)).to(DEV), (0.2 * rnd(Cw, seed=13)).to(DEV)
    Wq = bf(rnd(Cw, Cw, seed=21, scale=Cw ** -0.5))
    a = _ln_gemm(trunk, gamma, beta, Wq, None, rps=rps, xb=xb, part=part, bn=bn)
    assert torch.equal(a, _ln_gemm(trunk, gamma, beta, Wq, None, rps=rps, xb=xb, part=part, bn=bn))
    ref = F.layer_norm(trunk.float(), (Cw,), gamma, beta, 1e-5) @ Wq.float().t()
    easy = (~hard).to(DEV)
    report("folded to_q behind to_out, ordinary rows", a[easy], ref[easy], **BF16_OUT)
    e_hard = (a[hard.to(DEV)].float() - ref[hard.to(DEV)]).pow(2).mean().sqrt().item()
    e_easy = (a[easy].float() - ref[easy]).pow(2).mean().sqrt().item()
    print(f"  rms error: ordinary rows {e_easy:.3e}, rows with mean 30 sigma {e_hard:.3e}")
    assert e_hard < 45 * e_easy
    b = _ln_gemm(trunk, gamma, beta, Wq, None, rps=rps)                      # the stand-alone (xb, partials) of the same trunk
    report("same, (xb, partials) from the stand-alone kernel", b[easy], ref[easy], **BF16_OUT)


def test_shapes_without_a_folded_form_say_so():
    lib = _lib()
    x = _trunk(128, 640, seed=1)
    g, b = torch.ones(640, device=DEV), torch.zeros(640, device=DEV)
    W = bf(rnd(640, 640, seed=2))
    out = torch.empty(128, 640, device=DEV, dtype=torch.bfloat16)
    # 128 tokens per stream: the 16x16x32 family does not take the problem (small maps stay on gemm.hip) -> the engine keeps the LayerNorm launch
    assert lib.rt_op_ln_gemm(_ptr(x), _ptr(g), _ptr(b), _ptr(W), None, _ptr(out), 128, 640, 640, 0, 0, 128, None, None, 320, None) == -5
