#!/usr/bin/env python3
"""Stand-alone timing of the self-attention launch of an INJECTED rich-text step (config 3: 7 streams, q / k source [0, 1, 2, 3, 3, 3, 3])
in every mode of launch_attention_units (csrc/attention.hip; rt_op_gemm_debug bits 24 - 26), plus the same launch without injection.

  python tools/attn_units_bench.py            # the two SDXL attention levels: 1024 tokens x 20 heads, 4096 tokens x 10 heads
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, attention  # noqa: E402
from rich_text_to_image_amd.engine import load_library  # noqa: E402


def main():
    lib = load_library()
    for (H, N) in ((20, 1024), (10, 4096)):
        B, d = 7, 64
        g = torch.Generator().manual_seed(1)
        qs = d ** -0.5 * math.log2(math.e)
        Q = (torch.randn(B * N, H * d, generator=g) * qs).to(DEV).to(torch.bfloat16)
        K = torch.randn(B * N, H * d, generator=g).to(DEV).to(torch.bfloat16)
        VT = torch.randn(H * d, B * N, generator=g).to(DEV).to(torch.bfloat16)
        flops = 4.0 * B * H * N * N * d
        for name, src in (("plain   ", list(range(B))), ("injected", [0, 1, 2, 3, 3, 3, 3])):
            for mode in (1, 1, 2, 3, 4, 5):
                lib.rt_op_gemm_debug(mode << 24)
                for _ in range(3):
                    attention(Q, K, VT, B, H, N, N, d, q_src=src, k_src=src, v_src=list(range(B)))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                import ctypes as C
                from rich_text_to_image_amd.engine import _ptr
                O = torch.zeros(B * N, H * d, device=DEV, dtype=torch.bfloat16)
                ia = lambda v: (C.c_int * B)(*v)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    lib.rt_op_attention(_ptr(Q), Q.stride(0), _ptr(K), K.stride(0), _ptr(VT), VT.stride(0), _ptr(O), O.stride(0), ia(src), ia(src),
                                        ia(list(range(B))), None, None, None, B, H, N, N, N, d, 0, None)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                print(f"H {H:2d} N {N:4d} {name} mode {mode}: {us:8.1f} us   {flops / us * 1e-6:7.1f} TFLOP/s of the 7-stream algorithmic work", flush=True)
    lib.rt_op_gemm_debug(0)


if __name__ == "__main__":
    main()
