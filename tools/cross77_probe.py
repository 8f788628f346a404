"""cross77_kernel (csrc/cross77.hip) against attn_kernel<CROSS> (debug bit 19) through rt_op_attention at the two SDXL shapes - GPU box."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rich_text_to_image_amd.engine import _ptr, load_library  # noqa: E402

lib = load_library()
dev = "cuda:0"
for name, B, H, N in (("B 7 x 1024 x 20 heads", 7, 20, 1024), ("A 7 x 4096 x 10 heads", 7, 10, 4096), ("plain pass 2 x 1024 x 20", 2, 20, 1024)):
    HD, P = H * 64, 5
    g = torch.Generator(device=dev).manual_seed(1)
    bf = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
    Q, K, VT, O = bf(B * N, HD), bf(P * 96, HD), bf(HD, P * 96), torch.empty(B * N, HD, device=dev, dtype=torch.bfloat16)
    wabs = torch.zeros(2, 96, device=dev); wabs[:, :77] = 1.0
    wsgn = torch.ones(2, 96, device=dev)
    ia = lambda v: (C.c_int * B)(*v)
    src, prm, ws = ia(range(B)), ia([0, 4, 0, 4, 1, 2, 3][:B]), ia([-1, 1, -1, -1, -1, -1, -1][:B])
    res = {}
    for flags in (0, 1048576, 2097152, 3145728, 524288):
        lib.rt_op_gemm_debug(flags)
        def call():
            rc = lib.rt_op_attention(_ptr(Q), HD, _ptr(K), HD, _ptr(VT), P * 96, _ptr(O), HD, src, prm, prm, ws, _ptr(wabs), _ptr(wsgn), B, H, N, 96, 77, 64, 1, None)
            assert rc == 0, lib.rt_op_last_error().decode()
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            call()
        e1.record(); torch.cuda.synchronize()
        res[flags] = e0.elapsed_time(e1) / 50 * 1e3
    lib.rt_op_gemm_debug(0)
    mb = 2 * B * N * HD * 2 / 1e6
    print(f"{name}: cross77 {res[0]:.1f} us ({mb / res[0]:.2f} TB/s of Q + O), one tile per wave (bit 20) {res[1048576]:.1f} us, two heads per workgroup (bit 21) {res[2097152]:.1f} us, both {res[3145728]:.1f} us, attn_kernel<CROSS> {res[524288]:.1f} us")
