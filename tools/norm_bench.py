"""HBM-bound glue kernels at the SDXL config-3 shapes (7 streams): achieved bytes / time against the ~6.3 TB/s a streaming kernel
reaches on MI355X.  python tools/norm_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, groupnorm, layernorm  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    B = 7
    for rows, C in ((B * 1024, 1280), (B * 4096, 640)):
        x = torch.randn(rows, C, device=DEV); g = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
        dt = timeit(lambda: layernorm(x, g, b))
        print(f"layernorm {rows}x{C}: {dt * 1e6:7.1f} us  {rows * C * 6 / dt / 1e12:5.2f} TB/s (includes the wrapper's sync per call)")
    for HW, C1, C2, bf16in in ((1024, 1280, 0, False), (1024, 1280, 1280, False), (4096, 640, 0, False), (4096, 640, 640, False),
                               (16384, 320, 0, False), (1024, 1280, 0, True), (4096, 640, 0, True), (16384, 320, 0, True)):
        x1 = torch.randn(B, HW, C1, device=DEV)
        if bf16in:
            x1 = x1.to(torch.bfloat16)
        x2 = torch.randn(B, HW, C2, device=DEV) if C2 else None
        C = C1 + C2
        g = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
        dt = timeit(lambda: groupnorm(x1, x2, 32, g, b, 1e-5, True))
        nbytes = B * HW * C * ((2 if bf16in else 4) * 2 + 2)            # two reads + one bf16 write
        print(f"groupnorm HW={HW} C={C1}+{C2} bf16in={bf16in}: {dt * 1e6:7.1f} us  {nbytes / dt / 1e12:5.2f} TB/s (3 launches + wrapper sync)")


if __name__ == "__main__":
    main()
