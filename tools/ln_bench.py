"""LayerNorm of the fp16 trunk at the two SDXL shapes (7 streams), back-to-back launches timed with events.  python tools/ln_bench.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, _ptr, chk, load_library  # noqa: E402

lib = load_library()
for rows, Cc in ((7 * 1024, 1280), (7 * 4096, 640), (2 * 4096, 320)):
    x = torch.randn(rows, Cc, device=DEV).to(torch.float16)
    g, b = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    out = torch.empty(rows, Cc, device=DEV, dtype=torch.bfloat16)

    def run():
        chk(lib.rt_op_layernorm_f16(_ptr(x), _ptr(g), _ptr(b), _ptr(out), rows, Cc, C.c_float(1e-5), None))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f"layernorm fp16 -> bf16 {rows} x {Cc}: {us:6.1f} us  {rows * Cc * 4 / us / 1e6:5.2f} TB/s (read + write)")
