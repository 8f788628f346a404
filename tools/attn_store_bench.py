"""Stand-alone timing of the head-averaged map accumulation (rt_op_attention_probs_avg -> csrc/attn_store.hip) at the shapes the SDXL
token-map hooks record: 32x32 self maps (20 heads) and N x 77 cross maps; round-4 one-pass kernel against the round-1 two-pass kernel
(debug bit 5).   python tools/attn_store_bench.py   (GPU box, repo root)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, _ptr, chk, load_library  # noqa: E402

lib = load_library()
for name, H, N, NK, NKpad, ldq in (("self 1024x1024 (1280 ch)", 20, 1024, 1024, 1024, 2560), ("cross 1024x77 (1280 ch)", 20, 1024, 77, 96, 1280),
                                   ("cross 4096x77 (640 ch)", 10, 4096, 77, 96, 640)):
    DP = 64
    Q = (torch.randn(2 * N, ldq, device=DEV) * 0.5).bfloat16()
    K = Q[:, H * DP:] if ldq == 2 * H * DP else (torch.randn(8 * 96, H * DP, device=DEV) * 0.5).bfloat16()
    out = torch.zeros(N, NK, device=DEV)
    for legacy, label in ((0, "round 4 default       "), (64, "one-pass (round 4a)   "), (32, "two-pass (round 1)    ")):
        lib.rt_op_gemm_debug(legacy)

        def run():
            chk(lib.rt_op_attention_probs_avg(_ptr(Q), ldq, C.c_longlong(N), C.c_void_p(K.data_ptr()), K.stride(0), C.c_longlong(N if NK == N else 96), _ptr(out),
                                              H, N, NK, NKpad, NK if NK == N else 96, DP, 1, None))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        by = 8.0 * N * NK + 2.0 * (N + (NK if NK == N else 96)) * H * DP
        print(f"{name}: {label} {us:8.1f} us   {by / us / 1e6:.3f} TB/s of algorithmic bytes ({by / 1e6:.1f} MB)")
lib.rt_op_gemm_debug(0)
