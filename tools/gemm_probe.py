"""Runs a few GEMM shapes under fixed tile configurations (for rocprofv3 --pmc passes)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hiputil import DEV, bf  # noqa: E402
from rich_text_to_image_amd.engine import load_library, _ptr  # noqa: E402
lib = load_library()
shapes = [(7168, 1280, 1280, 2), (7168, 1280, 5120, 2), (7168, 10240, 1280, 3), (8192, 8192, 8192, 3), (7168, 1280, 1280, 0)]
for (M, N, K, cfg) in shapes:
    A = bf(torch.randn(M, K)); W = bf(torch.randn(N, K) * K ** -0.5)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    lib.rt_op_gemm_force_config(cfg)
    for _ in range(5):
        lib.rt_op_gemm(_ptr(A), _ptr(W), None, _ptr(out), None, None, 0, 0, M, N, K, K, K, N, 0, 0, 0, 0, 0, 0, 0, 0, None)
    torch.cuda.synchronize()
