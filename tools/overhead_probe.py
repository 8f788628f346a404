"""Fixed-cost probes for the GEMM kernel (launch + prologue + epilogue)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, bf  # noqa: E402
from rich_text_to_image_amd.engine import load_library, _ptr  # noqa: E402
lib = load_library()


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


x = torch.zeros(16, device=DEV)
print("tiny torch kernel (launch floor): %.1f us" % timeit(lambda: x.add_(1.0)))
for cfg in (2, 5, 6):            # 256x160 lock-step, 256x128 loader-wave, 256x160 ping-pong
    lib.rt_op_gemm_force_config(cfg)
    bn = 128 if cfg == 5 else 160
    for (M, N, K) in [(256, bn, 64), (256, bn, 2560), (7168, 1280, 2560), (7168, 1280, 1280)]:
        A = bf(torch.randn(M, K)); W = bf(torch.randn(N, K) * K ** -0.5)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)

        def f():
            lib.rt_op_gemm(_ptr(A), _ptr(W), None, _ptr(out), None, None, 0, 0, M, N, K, K, K, N, N, 0, 0, 0, 0, 0, 0, 0, None)
        print(f"cfg{cfg} {M}x{N}x{K}: {timeit(f):.1f} us")
