#!/usr/bin/env python3
"""SD-v1.5's small-M GEMMs / 3x3 convolutions (3 or 5 streams x 16^2 / 8^2 tokens) with and without the split-K path (rt_op_gemm_debug bit 2):
kernel time per call (events around 50 back-to-back calls).   python tools/small_gemm_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rich_text_to_image_amd.engine import load_library, _ptr  # noqa: E402

lib = load_library()
DEV = "cuda:0"


def t(M, N, K, epi=0, conv=None, flags=0, rps=0, res=False):
    g = torch.Generator().manual_seed(1)
    if conv is None:
        A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
        mode, Hin, Cin, Hout = 0, 0, 0, 0
        lda = K
    else:
        B, H, Cin = conv
        A = torch.randn(B, H, H, Cin, generator=g).to(DEV).to(torch.bfloat16)
        mode, Hin, Hout, lda, rps = 1, H, H, 0, H * H
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N // 2 if epi == 3 else N, device=DEV, dtype={4: torch.float16}.get(epi, torch.bfloat16))
    r = torch.randn(M, N, generator=g).to(DEV).to(torch.float16) if res else None
    lib.rt_op_gemm_debug(flags)

    def go():
        rc = lib.rt_op_gemm(_ptr(A), _ptr(W), _ptr(bias), _ptr(out), _ptr(r), None, mode, epi, M, N, K, lda, K, out.stride(0), r.stride(0) if r is not None else 0, 0,
                            rps, Hin, Hin, Cin, Hout, Hout, None)
        assert rc == 0, lib.rt_op_last_error().decode()
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            go()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    lib.rt_op_gemm_debug(0)
    return best


for streams in (3, 5):
    print(f"--- {streams} streams")
    for tokens in (64, 256):
        M = streams * tokens
        for (N, K, epi, res, what) in ((1280, 1280, 4, True, "to_out"), (1280, 1280, 0, False, "to_q"), (1280, 5120, 4, True, "ff.net.2"), (10240, 1280, 3, False, "GEGLU"), (2560, 1280, 0, False, "Q|K")):
            a, b = t(M, N, K, epi, rps=tokens, res=res), t(M, N, K, epi, flags=4, rps=tokens, res=res)
            print(f"dense {what:8s} {M:5d} x {N:5d} x {K:4d}: default {a:6.1f} us | no split-K {b:6.1f} us")
        for Cin in (1280, 2560):
            H = 8 if tokens == 64 else 16
            a, b = t(M, 1280, 9 * Cin, 0, conv=(streams, H, Cin)), t(M, 1280, 9 * Cin, 0, conv=(streams, H, Cin), flags=4)
            c = t(M, 1280, 9 * Cin, 0, conv=(streams, H, Cin), flags=1 << 28)      # bit 28: the split-K implicit GEMM of rounds 2 - 5 instead of the chunk-split patch kernel
            print(f"conv  {streams} x {H}x{H} x {Cin} -> 1280        : default {a:6.1f} us | no split-K {b:6.1f} us | split-K implicit GEMM {c:6.1f} us")
