"""The arithmetic of the VAE's precise mode (csrc/vae.hip) restated in numpy: operands as bf16 pairs hi = bf16(v),
lo = bf16(v - hi), products hi*hi + lo*hi + hi*lo accumulated in fp32.  Pins the error level the GPU tests then measure through the
whole decoder (1.8e-5): ~2^-16 per contraction against ~2^-9 for a single bf16 pass, on a contraction of a VAE layer's depth."""
import numpy as np


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def rel(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def test_three_pass_split_products_are_fp32_class():
    rng = np.random.default_rng(0)
    M, N, K = 96, 80, 1152                                  # K = 9 * 128: a 3x3 convolution over 128 channels
    A = rng.standard_normal((M, K)).astype(np.float32) * 1.7
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    Ah, Wh = bf16(A), bf16(W)
    Al, Wl = bf16(A - Ah), bf16(W - Wh)
    one = (Ah @ Wh.T).astype(np.float32)
    three = one + (Al @ Wh.T).astype(np.float32) + (Ah @ Wl.T).astype(np.float32)
    e1, e3 = rel(one, ref), rel(three, ref)
    print(f"single bf16 pass rel-L2 {e1:.2e}; hi*hi + lo*hi + hi*lo rel-L2 {e3:.2e}")
    assert 1e-3 < e1 < 6e-3                                 # ~2^-9 per operand
    assert e3 < 2e-5                                        # the dropped lo*lo term and the 16-bit pairs: ~2^-17 .. 2^-16
    # the pair itself carries 16 mantissa bits
    assert np.abs((Ah.astype(np.float64) + Al) - A).max() / np.abs(A).max() < 2.0 ** -16
