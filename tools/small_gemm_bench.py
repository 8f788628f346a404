import sys, os, torch, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from hiputil import gemm, DEV
from rich_text_to_image_amd.engine import load_library
lib = load_library()
def t(M, N, K, epi=0, mode=0, conv=None, flags=0, cfg=-1):
    g = torch.Generator().manual_seed(1)
    if mode == 0:
        A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    else:
        B, H, Cin = conv
        A = torch.randn(B, H, H, Cin, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(DEV)
    lib.rt_op_gemm_debug(flags); lib.rt_op_gemm_force_config(cfg)
    f = (lambda: gemm(A, W, bias, epi=epi)) if mode == 0 else (lambda: gemm(A, W, bias, epi=epi, mode=1, conv=(conv[1], conv[1])))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    lib.rt_op_gemm_debug(0); lib.rt_op_gemm_force_config(-1)
    return best
for (M, N, K) in ((192, 1280, 1280), (192, 1280, 5120), (192, 10240, 1280), (768, 1280, 1280), (768, 1280, 5120)):
    print(f"dense {M}x{N}x{K}: default {t(M,N,K):.1f} us | no split-K {t(M,N,K,flags=4):.1f} us (per call incl. python + sync overhead)")
for (B, H, Cin, Cout) in ((3, 8, 1280, 1280), (3, 8, 2560, 1280), (3, 16, 1280, 1280), (3, 16, 2560, 1280)):
    print(f"conv {B}x{H}x{H}x{Cin}->{Cout}: default {t(B*H*H, Cout, 9*Cin, mode=1, conv=(B,H,Cin)):.1f} us | no split-K {t(B*H*H, Cout, 9*Cin, mode=1, conv=(B,H,Cin), flags=4):.1f} us")
