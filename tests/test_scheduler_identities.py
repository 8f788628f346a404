"""The scheduler / embedding restatements that cannot be pinned against diffusers 0.18.2 here (it is not on disk) are at least
pinned against the MATHEMATICS they implement: the PNDM transfer formula is the deterministic DDIM update, the PLMS weights are
Adams-Bashforth, the Euler sigma-space step is the same DDIM update after the VP <-> VE change of variables, and the timestep
embedding is a unit-norm sinusoid basis.  (SURVEY 8c 'parity unpinned' rows; CPU only.)"""
import math

import torch

from oracle.schedulers import OracleEuler, OraclePNDM
from oracle.unet import timestep_embedding


def _ddim(x, eps, a_t, a_p):
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


def test_pndm_transfer_formula_is_deterministic_ddim():
    s = OraclePNDM()
    g = torch.Generator().manual_seed(0)
    x, eps = torch.randn(4, 8, generator=g, dtype=torch.float64), torch.randn(4, 8, generator=g, dtype=torch.float64)
    s.alphas_cumprod = s.alphas_cumprod.double()
    for t, p in ((981, 961), (501, 481), (21, 1), (1, -19)):
        a_t = float(s.alphas_cumprod[t]); a_p = float(s.alphas_cumprod[p]) if p >= 0 else float(s.final_alpha_cumprod)
        assert torch.allclose(s._prev(x, t, p, eps), _ddim(x, eps, a_t, a_p), rtol=1e-9, atol=1e-9), (t, p)


def test_plms_history_weights_are_adams_bashforth():
    """For a constant eps history every order must return eps itself (weights sum to 1); for a linear history the k-step
    Adams-Bashforth extrapolation is exact at the half step for order 2 (3/2, -1/2)."""
    # constant eps: each PLMS update must equal the plain DDIM update with that eps between the same (t, prev) pair
    s2 = OraclePNDM(); s2.set_timesteps(10)
    ts = s2.timesteps.tolist()
    ratio = 100
    x = torch.ones(3)
    for i, t in enumerate(ts[:7]):
        got = s2.step(torch.full((3,), 2.0), t, x)["prev_sample"]
        if i == 1:                                   # second call re-does the first step from the saved sample with averaged eps
            a_t, a_p = float(s2.alphas_cumprod[t + ratio]), float(s2.alphas_cumprod[t])
            src = torch.ones(3)
        else:
            a_t = float(s2.alphas_cumprod[t]); p = t - ratio
            a_p = float(s2.alphas_cumprod[p]) if p >= 0 else float(s2.final_alpha_cumprod)
            src = x
        assert torch.allclose(got, _ddim(src, torch.full((3,), 2.0), a_t, a_p), rtol=1e-5, atol=1e-6), i
        x = got
    for w in ([1.0], [1.5, -0.5], [23 / 12, -16 / 12, 5 / 12], [55 / 24, -59 / 24, 37 / 24, -9 / 24]):
        assert abs(sum(w) - 1) < 1e-12
        # exactness on polynomials of degree < order: integral over [0,1] of the interpolant through f(0), f(-1), ...
        for deg in range(len(w)):
            f = lambda u: u ** deg
            assert abs(sum(wi * f(-j) for j, wi in enumerate(w)) - 1.0 / (deg + 1)) < 1e-12, (w, deg)


def test_euler_step_is_ddim_in_sigma_space():
    e = OracleEuler(); e.set_timesteps(8)
    g = torch.Generator().manual_seed(1)
    x, eps = torch.randn(5, generator=g, dtype=torch.float64), torch.randn(5, generator=g, dtype=torch.float64)
    assert abs(e.init_noise_sigma - float((e.sigmas.max() ** 2 + 1) ** 0.5)) < 1e-6
    for i, t in enumerate(e.timesteps[:-1]):
        s, sn = float(e.sigmas[i]), float(e.sigmas[i + 1])
        got = e.step(eps, t, x)["prev_sample"]
        assert torch.allclose(got, (x - s * eps) + sn * eps, rtol=1e-6, atol=1e-6)          # stays on the ray x0 + sigma * eps
        # VE -> VP: x~ = x / sqrt(sigma^2 + 1), alpha_bar = 1 / (sigma^2 + 1); the DDIM update of x~ is the Euler update of x
        a, an = 1 / (s * s + 1), 1 / (sn * sn + 1)
        assert torch.allclose(_ddim(x * a ** 0.5, eps, a, an), got * an ** 0.5, rtol=1e-6, atol=1e-6)
        assert torch.allclose(e.scale_model_input(x.float(), t).double(), x * a ** 0.5, rtol=1e-5, atol=1e-6)
    last = e.step(eps, e.timesteps[-1], x)["prev_sample"]                                   # sigma_next = 0: the x0 prediction
    assert torch.allclose(last, x - float(e.sigmas[-2]) * eps, rtol=1e-6, atol=1e-6)


def test_timestep_embedding_is_a_unit_sinusoid_basis():
    t = torch.tensor([0.0, 1.0, 500.0, 999.0])
    for dim in (320, 256):
        emb = timestep_embedding(t, dim)
        half = dim // 2
        assert torch.allclose(emb[:, :half] ** 2 + emb[:, half:] ** 2, torch.ones(4, half), atol=1e-5)       # cos^2 + sin^2
        assert torch.allclose(emb[0, :half], torch.ones(half)) and torch.allclose(emb[0, half:], torch.zeros(half))  # [cos | sin] at t = 0
        freq = torch.exp(-math.log(10000.0) * torch.arange(half) / half)
        assert torch.allclose(emb[1, half:], torch.sin(freq), atol=1e-6) and abs(float(freq[0]) - 1.0) < 1e-7
