"""GroupNorm + SiLU on the fp16 trunk at the SD-v1.5 / SDXL shapes: the one-launch form against the two-launch form (rt_op_gemm_debug bit 23); 50 back-to-back calls
through ctypes, so the smallest shapes read the call overhead, not the kernel.  python tools/gn_one_launch_bench.py"""
import os, sys, ctypes as C, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from rich_text_to_image_amd.engine import load_library, _ptr
lib = load_library(); DEV = "cuda:0"
def t(B, HW, C1, C2, flags):
    x1 = torch.randn(B, HW, C1, device=DEV).half(); x2 = torch.randn(B, HW, C2, device=DEV).half() if C2 else None
    Cc = C1 + C2; g = torch.ones(Cc, device=DEV); b = torch.zeros(Cc, device=DEV); out = torch.empty(B, HW, Cc, device=DEV, dtype=torch.bfloat16)
    lib.rt_op_gemm_debug(flags)
    def go():
        rc = lib.rt_op_groupnorm(_ptr(x1), _ptr(x2), 2, C1, C2, 32, B, HW, _ptr(g), _ptr(b), C.c_float(1e-5), 1, _ptr(out), None, None); assert rc == 0
    for _ in range(5): go()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(50): go()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    lib.rt_op_gemm_debug(0); return best
for (B, HW, C1, C2) in ((3, 256, 1280, 0), (3, 256, 1280, 1280), (3, 256, 1280, 640), (3, 1024, 640, 0), (3, 1024, 640, 640), (5, 256, 1280, 1280), (5, 1024, 640, 640),
                        (3, 4096, 320, 320), (7, 1024, 1280, 0), (7, 1024, 1280, 1280), (7, 1024, 1280, 640), (7, 4096, 640, 0), (7, 4096, 640, 640), (2, 1024, 1280, 0), (2, 4096, 640, 0)):
    print(f"groupnorm+silu fp16 B={B} HW={HW} C={C1}+{C2}: one launch {t(B, HW, C1, C2, 0):6.1f} us | two launches {t(B, HW, C1, C2, 1 << 23):6.1f} us", flush=True)
