"""GroupNorm (two launches + wrapper) on the VAE-sized fp32 maps and the UNet-sized fp16 maps: achieved bytes / time.  python tools/gn_big_bench.py"""
import os, sys, ctypes as C, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from rich_text_to_image_amd.engine import load_library, _ptr
lib = load_library(); DEV = "cuda:0"
def t(B, HW, C1, dt, G=32):
    x1 = torch.randn(B, HW, C1, device=DEV).to(dt)
    g = torch.ones(C1, device=DEV); b = torch.zeros(C1, device=DEV); out = torch.empty(B, HW, C1, device=DEV, dtype=torch.bfloat16)
    it = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dt]
    def go():
        rc = lib.rt_op_groupnorm(_ptr(x1), None, it, C1, 0, G, B, HW, _ptr(g), _ptr(b), C.c_float(1e-5), 1, _ptr(out), None, None); assert rc == 0
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nbytes = B * HW * C1 * (x1.element_size() * 2 + 2)
    print(f"groupnorm B={B} HW={HW} C={C1} {dt}: {us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s (2 reads + 1 bf16 write)", flush=True)
t(1, 1 << 20, 128, torch.float32); t(1, 1 << 18, 256, torch.float32); t(1, 1 << 16, 512, torch.float32)
t(7, 16384, 320, torch.float16); t(7, 4096, 640, torch.float16)
