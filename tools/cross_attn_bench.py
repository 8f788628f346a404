"""Stand-alone timing of the 96-key cross-attention kernel through rt_op_attention at the two SDXL shapes: multiplier tables (wset 0)
against the plain path (wset -1).   python tools/cross_attn_bench.py   (GPU box, repo root)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, _ptr, chk, load_library  # noqa: E402

lib = load_library()
for B, H, N in ((7, 20, 1024), (7, 10, 4096)):
    DP, P = 64, 8
    Q = torch.randn(B * N, H * DP, device=DEV).bfloat16()
    K = torch.randn(P * 96, H * DP, device=DEV).bfloat16()
    VT = torch.randn(H * DP, P * 96, device=DEV).bfloat16()
    O = torch.empty_like(Q)
    wabs = torch.ones(2, 96, device=DEV); wabs[:, 77:] = 0
    wsgn = torch.ones(2, 96, device=DEV)
    src = (C.c_int * B)(*range(B))
    ksrc = (C.c_int * B)(*[b % P for b in range(B)])
    for name, ws in (("tables", [0] * B), ("plain ", [-1] * B), ("mixed ", [-1, 1] + [-1] * (B - 2))):
        wset = (C.c_int * B)(*ws)

        def run():
            chk(lib.rt_op_attention(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(VT), VT.stride(0), _ptr(O), O.stride(0), src, ksrc, ksrc,
                                    wset, _ptr(wabs), _ptr(wsgn), B, H, N, 96, 77, DP, 1, None))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"cross-attn B{B} H{H} N{N} {name}: {us:7.1f} us  ({2 * Q.numel() * 2 / us / 1e6:.2f} TB/s of Q + O)")
