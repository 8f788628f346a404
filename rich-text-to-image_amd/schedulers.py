"""Host-side scheduler tables for the engine (product code; the oracle keeps an independent copy).

Restates the published diffusers 0.18.2 `PNDMScheduler` (skip_prk_steps=True, steps_offset=1; used at
models/region_diffusion.py:35-37) and `EulerDiscreteScheduler` (SDXL config; models/region_diffusion_sdxl.py:120).
Only the *tables* live here (timesteps, sigmas, alphas_cumprod); the update arithmetic runs in
csrc/step.hip.  diffusers is third-party and not on disk => [memory], parity unpinned."""
import numpy as np
import torch


def alphas_cumprod(num_train=1000, beta_start=0.00085, beta_end=0.012):
    # fp32 torch arithmetic, as diffusers builds the "scaled_linear" schedule
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).numpy()


class PNDMTables:
    kind = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train=1000):
        self.num_train = num_train
        self.alphas_cumprod = alphas_cumprod(num_train)

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (self.num_train // n)).round() + 1
        self.timesteps = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].astype(np.int64).copy()
        return self

    def table(self):
        return self.alphas_cumprod.tolist()


class EulerTables:
    kind = 0

    def __init__(self, num_train=1000):
        self.num_train = num_train
        ac = alphas_cumprod(num_train).astype(np.float64)
        self.alphas_cumprod = alphas_cumprod(num_train)
        self._train_sigmas = ((1 - ac) / ac) ** 0.5
        self.init_noise_sigma = float(self._train_sigmas.max())

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ts = (np.arange(0, n) * (self.num_train // n)).round()[::-1].copy().astype(np.float32) + 1
        sig = np.interp(ts, np.arange(0, self.num_train), self._train_sigmas)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = ts
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)
        return self

    def table(self):
        return self.sigmas.tolist()
