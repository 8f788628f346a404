"""Kernel micro-benchmarks on the GPU box (GEMM / conv / attention TF/s at the SDXL shapes, one batched UNet forward)."""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, attention, bf, gemm  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    res = {}
    B = 7
    for (M, N, K, name) in [(B * 1024, 1280, 1280, "to_q/out 1024"), (B * 1024, 10240, 1280, "geglu 1024"),
                            (B * 1024, 1280, 5120, "ff2 1024"), (B * 4096, 640, 640, "to_q/out 4096"),
                            (B * 4096, 5120, 640, "geglu 4096"), (B * 4096, 640, 2560, "ff2 4096"), (B * 1024, 2560, 1280, "qk 1024"), (B * 1024, 1280, 64, "K=64 overhead probe"), (B * 1024, 1280, 256, "K=256 overhead probe"), (1280, B * 1024, 1280, "vt 1024"), (8192, 8192, 8192, "8k cube")]:
        A = bf(torch.randn(M, K)); W = bf(torch.randn(N, K) * K ** -0.5)
        import ctypes as C
        from rich_text_to_image_amd.engine import load_library, _ptr
        lib = load_library()
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)

        def f():
            lib.rt_op_gemm(_ptr(A), _ptr(W), None, _ptr(out), None, None, 0, 0, M, N, K, K, K, N, 0, 0, 0, 0, 0, 0, 0, 0, None)
        per = {}
        for cfg in (0, 2, 3, 4, 5, -1):
            lib.rt_op_gemm_force_config(cfg)
            dt = timeit(f)
            per["auto" if cfg < 0 else f"cfg{cfg}"] = round(2 * M * N * K / dt / 1e12, 1)
        res[f"gemm {name} {M}x{N}x{K}"] = per
    for (Bc, H, W_, Cin, Cout, name) in [(B, 128, 128, 320, 320, "conv 128^2 320"), (B, 64, 64, 640, 640, "conv 64^2 640"),
                                         (B, 32, 32, 1280, 1280, "conv 32^2 1280"), (B, 32, 32, 2560, 1280, "conv 32^2 2560->1280")]:
        A = bf(torch.randn(Bc, H, W_, Cin)); Wt = bf(torch.randn(Cout, 9 * Cin) * (9 * Cin) ** -0.5)
        out = torch.empty(Bc * H * W_, Cout, device=DEV, dtype=torch.float32)
        M = Bc * H * W_

        def f():
            lib.rt_op_gemm(_ptr(A), _ptr(Wt), None, _ptr(out), None, None, 1, 1, M, Cout, 9 * Cin, 0, 9 * Cin, Cout, 0, 0, H * W_, H, W_, Cin, H, W_, None)
        per = {}
        for cfg in (0, 2, 3, 4, 5, -1):
            lib.rt_op_gemm_force_config(cfg)
            dt = timeit(f)
            per["auto" if cfg < 0 else f"cfg{cfg}"] = round(2 * M * Cout * 9 * Cin / dt / 1e12, 1)
        res[name] = per
    for (H, N, name) in [(10, 4096, "self-attn 4096 h10"), (20, 1024, "self-attn 1024 h20")]:
        DP = 64
        Q = bf(torch.randn(B * N, H * DP) * 0.2); K = bf(torch.randn(B * N, H * DP)); VT = bf(torch.randn(H * DP, B * N))
        dt = timeit(lambda: attention(Q, K, VT, B, H, N, N, DP), iters=10)
        res[name] = dict(ms=dt * 1e3, tflops=4 * B * H * N * N * DP / dt / 1e12)
    for k, v in res.items():
        print(k, v)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w"), indent=1)

    # one batched SDXL forward (random weights)
    from rich_text_to_image_amd.engine import Engine, SDXL_CONFIG
    t0 = time.time()
    eng = Engine(SDXL_CONFIG, 128, 128, device=0, max_streams=8, max_prompts=8)
    eng.init_random_weights(0)
    print("engine create + random weights: %.1f s" % (time.time() - t0))
    P = 5
    eng.set_prompts(torch.randn(P, 77, 2048, device=DEV), torch.randn(P, 1280, device=DEV), torch.tensor([[1024., 1024, 0, 0, 1024, 1024]]))
    eng.set_fontsize(torch.tensor([5, 6]), torch.tensor([20.0, 20.0]))
    x = torch.randn(7, 4, 128, 128, device=DEV)
    for Bn in (1, 2, 7):
        fn = lambda: eng.unet_forward(x[:Bn], 801.0, [0, 4, 0, 4, 1, 2, 3][:Bn], fontsize=[0, 1, 0, 0, 0, 0, 0][:Bn])
        out = fn()
        print("SDXL fwd B=%d finite=%s std=%.3f" % (Bn, bool(torch.isfinite(out).all()), out.std().item()))
        dt = timeit(fn, iters=3, warm=1)
        res[f"sdxl_forward_B{Bn}"] = dict(ms=dt * 1e3, tflops=Bn * 6.7612 / dt)
        print(f"SDXL batched forward B={Bn}: {dt*1e3:.1f} ms  ({Bn * 6.7612 / dt:.1f} TFLOP/s)")
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "microbench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
