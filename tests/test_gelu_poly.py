"""The GEGLU epilogues evaluate gelu through a clamped odd polynomial (csrc/common.h::gelu_erf / gelu_erf_x2) instead of libm's erff.
This restates the device evaluation in numpy with the coefficients parsed out of common.h and pins its error against the exact
erf-gelu of the reference (diffusers GEGLU: F.gelu, models/attention.py:372-380): <= 7e-5 absolute, exact saturation outside the
clamp.  The bf16 output grid is coarser than that for |gelu| > 0.03."""
import os
import re
import sys

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _table():
    src = open(os.path.join(ROOT, "rich-text-to-image_amd", "csrc", "common.h")).read()
    c = float(re.search(r"#define RT_GELU_CLAMP ([0-9.]+)f", src).group(1))
    body = src[src.index("#define RT_GELU_COEFS(K)"):]
    body = body[:body.index("__device__")]
    lead = float(re.search(r"#define RT_GELU_C8 ([-+0-9.e]+)f", src).group(1))
    return c, [lead] + [float(v) for v in re.findall(r"K\(([-+0-9.e]+)f\)", body)]


def test_device_gelu_polynomial_against_exact_erf_gelu():
    from fit_gelu import gelu_poly
    c, co = _table()
    assert len(co) == 9 and c == 4.25
    x = np.linspace(-12, 12, 1000001)
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    got = gelu_poly(x, co, c).astype(np.float64)
    err = np.abs(got - ref)
    print(f"gelu polynomial: max abs error {err.max():.3e} at x = {x[err.argmax()]:.3f}")
    assert err.max() < 7e-5
    big = np.abs(x) >= c
    pos, neg = big & (x > 0), big & (x < 0)                               # saturated: x (1 + P(c)) / 2 with |P(c) - 1| ~ 1 fp32 ulp of the sum
    assert (np.abs(got[pos] - x[pos]) / x[pos]).max() < 4e-6 and (np.abs(got[neg]) / -x[neg]).max() < 4e-6
    # relative error where the result is not tiny
    m = np.abs(ref) > 0.03
    assert (err[m] / np.abs(ref[m])).max() < 2.5e-3             # bf16 half ulp: 2.0e-3 .. 3.9e-3


def test_committed_table_is_what_the_fit_script_produces():
    from fit_gelu import fit
    c, co = _table()
    again = fit(c)[::-1]
    assert np.allclose(np.array(co), again.astype(np.float32), rtol=2e-3, atol=1e-12)
