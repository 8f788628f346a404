"""Coefficients of csrc/common.h::gelu_erf: erf(x / sqrt 2) ~= P(clamp(x, +-c)), P(x) = x * Q(x^2) odd with P(c) == 1 exactly, by an
iteratively re-weighted least-squares (minimax-like) fit on Chebyshev nodes.  `python tools/fit_gelu.py` prints the table
(highest power first, as RT_GELU_COEFS lists it) and the worst absolute gelu error of the fp32 Horner evaluation on [-12, 12]."""
import numpy as np
from scipy.special import erf


def fit(c=4.25, deg=8, iters=200, n=8000):
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 * c + 0.5 * c
    t, tgt = (x / c) ** 2, erf(x / np.sqrt(2))
    w = np.ones(n)
    for _ in range(iters):
        # P(x) = x/c + sum_{k>=1} b_k x (t^k - 1): the constraint P(c) = 1 is built in
        A = np.stack([x * (t ** k - 1.0) for k in range(1, deg + 1)], 1)
        b = tgt - x / c
        co = np.linalg.lstsq(A * w[:, None], b * w, rcond=None)[0]
        e = A @ co - b
        w = w * (1 + 2 * np.abs(e) / np.abs(e).max())
        w /= w.mean()
    bb = np.concatenate([[1 / c - co.sum()], co])
    return bb / np.array([c ** (2 * k) for k in range(deg + 1)])          # coefficients of s = x^2, lowest power first


def gelu_poly(x, coefs_high_first, c):
    """the device evaluation restated in numpy (fp32 storage, fma emulated through fp64)"""
    x = np.asarray(x, np.float32)
    xc = np.clip(x, -np.float32(c), np.float32(c))
    s = xc * xc
    q = np.zeros_like(x)
    for k in coefs_high_first:
        q = (q.astype(np.float64) * s + np.float32(k)).astype(np.float32)
    pe = xc * q
    h = x * np.float32(0.5)
    return (h.astype(np.float64) * pe + h).astype(np.float32)


if __name__ == "__main__":
    c = 4.25
    co = fit(c)[::-1]
    print(" ".join("K(%.9ef)" % float(np.float32(v)) for v in co))
    x = np.linspace(-12, 12, 4000001)
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    print("max |gelu error| %.3e" % np.abs(gelu_poly(x, co, c) - ref).max())
