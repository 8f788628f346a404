"""ORACLE (test infrastructure): schedulers restated from the published diffusers 0.18.2 algorithms.

The reference calls them at models/region_diffusion.py:35-37,95,139,147,177 (PNDMScheduler with
skip_prk_steps=True, steps_offset=1, scaled_linear betas 0.00085..0.012) and
models/region_diffusion_sdxl.py:120,536,735,771,784,799,837,845,956 (EulerDiscreteScheduler, SDXL
config: scaled_linear, steps_offset=1, timestep_spacing "leading").

diffusers==0.18.2 (environment.yaml:17) is NOT on disk: `scheduling_pndm.py` and
`scheduling_euler_discrete.py` are restated from memory of the published source => PARITY UNPINNED
for these two classes (nothing in /root/reference tests them).
The objects expose exactly the attribute/method surface the reference loops touch, so the
unmodified reference loops can be driven with them (oracle/make_golden.py).
"""
import numpy as np
import torch


def scaled_linear_alphas_cumprod(num_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class OraclePNDM:
    """PLMS branch of PNDMScheduler (skip_prk_steps=True, steps_offset=1)."""
    order = 1

    def __init__(self, num_train=1000):
        self.num_train = num_train
        self.alphas_cumprod = scaled_linear_alphas_cumprod(num_train)
        self.final_alpha_cumprod = self.alphas_cumprod[0]      # set_alpha_to_one=False
        self.init_noise_sigma = 1.0
        self.steps_offset = 1

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.num_train // n
        ts = (np.arange(0, n) * ratio).round() + self.steps_offset
        # skip_prk_steps: plms timesteps duplicate the second entry
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def scale_model_input(self, sample, t=None):
        return sample

    def _prev(self, sample, t, prev_t, eps):
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * eps / denom

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        t = int(timestep)
        prev_t = t - self.num_train // self.num_inference_steps
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = t
            t = t + self.num_train // self.num_inference_steps
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev = self._prev(sample, t, prev_t, model_output)
        self.counter += 1
        return {"prev_sample": prev} if return_dict else (prev,)


class OracleEuler:
    """EulerDiscreteScheduler, SDXL config (leading spacing, steps_offset 1, epsilon prediction)."""
    order = 1

    def __init__(self, num_train=1000):
        self.num_train = num_train
        self.alphas_cumprod = scaled_linear_alphas_cumprod(num_train)
        sig = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self._train_sigmas = sig.numpy().astype(np.float64)
        self.init_noise_sigma = float(max(self._train_sigmas))   # 0.18.2: sigmas.max() before set_timesteps
        self.steps_offset = 1

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.num_train // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.float32) + self.steps_offset
        sig = np.interp(ts, np.arange(0, self.num_train), self._train_sigmas)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts)
        # 0.18.2 (leading spacing): init_noise_sigma = (sigma_max**2 + 1) ** 0.5
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)

    def _index(self, t):
        return int((self.timesteps == float(t)).nonzero()[0].item())

    def scale_model_input(self, sample, t):
        s = self.sigmas[self._index(t)]
        return sample / ((s ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        i = self._index(timestep)
        s, s_next = self.sigmas[i], self.sigmas[i + 1]
        # epsilon prediction, gamma = 0: pred_x0 = x - s*eps; d = (x - pred_x0)/s = eps; x += d*(s_next - s)
        pred_original = sample - s * model_output
        derivative = (sample - pred_original) / s
        prev = sample + derivative * (s_next - s)
        return {"prev_sample": prev} if return_dict else (prev,)
