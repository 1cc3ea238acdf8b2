"""ORACLE (test infrastructure) — numpy restatement of diffusers 0.27.2 PNDMScheduler in the configuration the
reference builds at src/stable_diffusion_depth.py:98-100
(beta_start=0.00085, beta_end=0.012, "scaled_linear", 1000 train steps, steps_offset=1, skip_prk_steps=True)
and of DDPM add_noise (stable_diffusion_depth.py:364).  diffusers is not installed and the reference holds no
scheduler fixtures => PARITY UNPINNED against diffusers itself; the algorithm follows SURVEY.md Appendix A.4.
"""
import numpy as np


class PNDMRef:
    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000, steps_offset=1):
        self.T = num_train_timesteps
        self.betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod((1.0 - self.betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one=False
        self.steps_offset = steps_offset

    def set_timesteps(self, n):
        self.n = n
        ratio = self.T // n
        ts = (np.arange(0, n) * ratio).round() + self.steps_offset
        self.timesteps = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].astype(np.int64).copy()
        self.ets, self.counter, self.cur_sample = [], 0, None
        return self.timesteps

    def _prev(self, sample, t, prev_t, eps):
        a = self.alphas_cumprod[t]
        ap = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b, bp = np.float32(1) - a, np.float32(1) - ap
        sc = np.float32((ap / a) ** 0.5)
        den = a * np.float32(bp ** 0.5) + np.float32((a * b * ap) ** 0.5)
        return sc * sample - (ap - a) * eps / den

    def step(self, eps, t, sample):
        t = int(t)
        prev_t = t - self.T // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:] + [eps]
        else:
            prev_t, t = t, t + self.T // self.n
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            out = eps; self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            out = (eps + e[-1]) / 2; sample = self.cur_sample; self.cur_sample = None
        elif len(e) == 2:
            out = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            out = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            out = (1 / 24) * (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4])
        self.counter += 1
        return self._prev(sample, t, prev_t, out.astype(np.float32)).astype(np.float32)

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod[int(t)]
        return (np.float32(a ** 0.5) * x0 + np.float32((1 - a) ** 0.5) * noise).astype(np.float32)


def cfg(eps_pair, guidance):
    u, c = eps_pair[0], eps_pair[1]
    return u + np.float32(guidance) * (c - u)


class EulerAncestralRef:
    """numpy restatement of diffusers 0.27.2 EulerAncestralDiscreteScheduler (scaled_linear betas, linspace spacing or an explicit
    timestep list, epsilon / v prediction).  PARITY UNPINNED (diffusers absent offline): the formulas are the published ones —
    sigma = sqrt((1-abar)/abar); x_in = x / sqrt(sigma^2+1); x0 = x - sigma*eps  |  x0 = -sigma/sqrt(sigma^2+1) * v + x/(sigma^2+1);
    sigma_up = sqrt(s_to^2 (s^2 - s_to^2) / s^2); sigma_down = sqrt(s_to^2 - sigma_up^2); x' = x + (x - x0)/s * (sigma_down - s) + n*sigma_up."""

    def __init__(self, T=1000, beta_start=0.00085, beta_end=0.012, prediction_type="v_prediction"):
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float32) ** 2
        ac = np.cumprod(1.0 - betas)
        self.sig_all = np.sqrt((1 - ac) / ac).astype(np.float64)
        self.T, self.prediction_type = T, prediction_type

    def set_timesteps(self, n=None, timesteps=None):
        ts = np.linspace(0, self.T - 1, n)[::-1].copy() if timesteps is None else np.asarray(timesteps, np.float64)
        self.timesteps = ts
        self.sigmas = np.concatenate([np.interp(ts, np.arange(self.T), self.sig_all), [0.0]])
        return ts

    def scale(self, x, i):
        return x / np.sqrt(self.sigmas[i] ** 2 + 1)

    def step(self, out, i, x, noise):
        s, s_to = self.sigmas[i], self.sigmas[i + 1]
        x0 = x - s * out if self.prediction_type == "epsilon" else out * (-s / np.sqrt(s ** 2 + 1)) + x / (s ** 2 + 1)
        up = np.sqrt(s_to ** 2 * (s ** 2 - s_to ** 2) / s ** 2)
        down = np.sqrt(s_to ** 2 - up ** 2)
        return x + (x - x0) / s * (down - s) + noise * up
