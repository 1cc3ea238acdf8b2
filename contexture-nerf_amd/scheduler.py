"""PNDMScheduler: the diffusers seam used at src/stable_diffusion_depth.py:98-100,298,310,364,514
(`set_timesteps`, `.timesteps`, `.step(eps, t, x)['prev_sample']`, `.add_noise`, `.scale_model_input`,
`.alphas_cumprod`), PLMS-only (`skip_prk_steps=True`), epsilon prediction.

`step()` keeps the diffusers call shape.  `step_cfg()` additionally fuses the classifier-free-guidance
combine (stable_diffusion_depth.py:428-430) with the multistep update in one HIP kernel that reads the
[2,...] noise prediction once; the epsilon history lives in a 4-slot device ring.
"""
import ctypes as C
import numpy as np
import torch
from . import _lib as L


class PNDMScheduler:
    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                 steps_offset=1, skip_prk_steps=True, **kw):
        if beta_schedule != "scaled_linear" or not skip_prk_steps:
            raise L.CtxError("PNDMScheduler: only the reference's configuration (scaled_linear, skip_prk_steps) is implemented")
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self._ac = self.alphas_cumprod.numpy()
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None
        self._ring = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round() + self.steps_offset
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        self.counter = 0
        self._n_ets = 0
        self._head = -1
        self._ring = None
        self._cur = None

    def scale_model_input(self, sample, *a, **k):
        return sample

    def add_noise(self, original_samples, noise, timesteps):
        t = timesteps.reshape(-1).long().cpu()
        a = self.alphas_cumprod[t].to(original_samples.device, original_samples.dtype)
        sa, sb = (a ** 0.5).flatten(), ((1 - a) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise

    # -- coefficient bookkeeping shared by step() and step_cfg() ------------------------------------------------
    def _coeffs(self, t, prev_t):
        a = self._ac[t]
        ap = self._ac[prev_t] if prev_t >= 0 else self._ac[0]
        b, bp = np.float32(1) - a, np.float32(1) - ap
        sample_coeff = np.float32((ap / a) ** 0.5)
        den = a * np.float32(bp ** 0.5) + np.float32((a * b * ap) ** 0.5)
        return float(sample_coeff), float((ap - a) / den)

    def _advance(self, timestep):
        """-> (mode, head, coef4, t_eff, prev_t) and updates counters exactly like diffusers step_plms."""
        t = int(timestep)
        ratio = self.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self._n_ets = min(self._n_ets + 1, 4)
            self._head = (self._head + 1) & 3
            mode = 2 if self.counter == 0 else 0
        else:
            prev_t, t = t, t + ratio
            mode = 1
        n = self._n_ets
        if mode == 1 or n == 1:
            coef = [1.0, 0.0, 0.0, 0.0]
        elif n == 2:
            coef = [1.5, -0.5, 0.0, 0.0]
        elif n == 3:
            coef = [23 / 12, -16 / 12, 5 / 12, 0.0]
        else:
            coef = [55 / 24, -59 / 24, 37 / 24, -9 / 24]
        self.counter += 1
        return mode, self._head, coef, t, prev_t

    def _buffers(self, sample):
        n = sample.numel()
        if self._ring is None or self._ring.shape[1] != n or self._ring.device != sample.device:
            self._ring = torch.zeros(4, n, device=sample.device)
            self._cur = torch.empty(n, device=sample.device)

    def step_cfg(self, noise_pred_pair, guidance_scale, timestep, sample):
        """noise_pred_pair [2,...] = (uncond, text) -> {'prev_sample'}; fused CFG + PLMS (HIP)."""
        lib = L.load()
        x = L.f32c(sample).clone()
        pair = L.f32c(noise_pred_pair)
        if pair.numel() != 2 * x.numel():
            raise L.CtxError("step_cfg: noise_pred_pair must stack (uncond, text) along dim 0")
        self._buffers(x)
        mode, head, coef, t, prev_t = self._advance(timestep)
        sc, ec = self._coeffs(t, prev_t)
        c4 = (C.c_float * 4)(*coef)
        L.check(lib.ctx_cfg_plms_step(L.ptr(pair, torch.float32, "noise_pred"), x.numel(), float(guidance_scale), L.ptr(self._ring),
                                      head, c4, sc, ec, mode, L.ptr(self._cur), L.ptr(x), L.stream()))
        return {'prev_sample': x}

    def step(self, model_output, timestep, sample, **kw):
        """diffusers call shape (already-guided epsilon)."""
        pair = torch.stack([model_output, model_output])
        return self.step_cfg(pair, 0.0, timestep, sample)


class DDPMScheduler:
    """diffusers 0.27.2 DDPMScheduler, the subset the reference's Zero123++ path uses: `add_noise` (src/training/trainer.py:746)
    and, because `init_zero123plus` swaps it in as the PIPELINE's scheduler (`DDPMScheduler.from_config(pipeline.scheduler.config)`,
    trainer.py:306), `set_timesteps(timesteps=[t])`, the identity `scale_model_input`, `init_noise_sigma == 1` and the ancestral
    `step` (variance_type "fixed_small") for epsilon / v prediction.  `clip_sample` / `clip_sample_range` default to True / 1.0:
    the EulerAncestral config the reference builds this scheduler from carries no `clip_sample` key, so `from_config` yields the
    DDPM defaults and `pred_original_sample` is clamped to [-1, 1] before `prev_sample` is formed (the SDS loop only reads
    `noise_pred` through the callback and is unaffected; sampling through the pipeline is).  Host-side tensor arithmetic.
    PARITY UNPINNED vs diffusers (absent offline)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", clip_sample=True, clip_sample_range=1.0, **kw):
        self.clip_sample, self.clip_sample_range = bool(clip_sample), float(clip_sample_range)
        if beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise ValueError(f"DDPMScheduler: beta_schedule {beta_schedule!r} not implemented")
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self.custom_timesteps = False
        self.num_inference_steps = None

    add_noise = PNDMScheduler.add_noise
    scale_model_input = PNDMScheduler.scale_model_input

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        if timesteps is not None:
            ts = [int(round(float(t))) for t in timesteps]
            if any(ts[i] <= ts[i + 1] for i in range(len(ts) - 1)):
                raise ValueError("`custom_timesteps` must be in descending order.")
            if ts[0] >= self.num_train_timesteps:
                raise ValueError(f"`timesteps` must start before `self.config.train_timesteps`: {self.num_train_timesteps}.")
            self.timesteps, self.custom_timesteps = torch.tensor(ts, dtype=torch.int64), True
            self.num_inference_steps = len(ts)
        else:
            ratio = self.num_train_timesteps // num_inference_steps             # timestep_spacing "leading", steps_offset 0
            self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))
            self.custom_timesteps, self.num_inference_steps = False, num_inference_steps

    def previous_timestep(self, t):
        if self.custom_timesteps:
            idx = (self.timesteps == int(t)).nonzero()[0][0].item()
            return -1 if idx == len(self.timesteps) - 1 else int(self.timesteps[idx + 1])
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        return int(t) - self.num_train_timesteps // n

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        t = int(round(float(timestep.reshape(-1)[0]))) if isinstance(timestep, torch.Tensor) else int(round(float(timestep)))
        prev_t = self.previous_timestep(t)
        ac = self.alphas_cumprod
        a_t = float(ac[t]); a_prev = float(ac[prev_t]) if prev_t >= 0 else 1.0
        b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
        cur_a = a_t / a_prev; cur_b = 1.0 - cur_a
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
        else:
            raise ValueError(f"DDPMScheduler: prediction_type {self.prediction_type!r}")
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        prev = (a_prev ** 0.5 * cur_b / b_t) * x0 + (cur_a ** 0.5 * b_prev / b_t) * sample
        if t > 0:
            var = max(b_prev / b_t * cur_b, 1e-20)                                # fixed_small
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + var ** 0.5 * noise
        return {'prev_sample': prev, 'pred_original_sample': x0}


class EulerAncestralDiscreteScheduler:
    """diffusers 0.27.2 EulerAncestralDiscreteScheduler as Zero123++ samples with it (`val_sched` of RefOnlyNoisedUNet and the
    pipeline's scheduler, src/zero123plus.py:164-237, 411-746): sigma_t = sqrt((1 - abar_t) / abar_t), model input scaled by
    1 / sqrt(sigma^2 + 1), epsilon or v prediction, ancestral step with fresh noise.  Host-side tensor arithmetic only.
    PARITY UNPINNED vs diffusers (absent offline); oracle/scheduler.py restates the same published algorithm in numpy.
    `set_timesteps(timesteps=[...])` takes an explicit schedule (the SDS loop denoises one step at its DreamTime t,
    src/training/trainer.py:785-795)."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="v_prediction", timestep_spacing="linspace", steps_offset=0, **kw):
        if beta_schedule != "scaled_linear":
            raise ValueError(f"beta_schedule {beta_schedule!r} is not implemented")
        self.num_train_timesteps, self.prediction_type = num_train_timesteps, prediction_type
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self._sig_all = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.set_timesteps(num_train_timesteps)

    def _sigma_at(self, t):
        """linear interpolation of the training sigmas at (fractional) timesteps t."""
        t = torch.as_tensor(t, dtype=torch.float64)
        lo = t.floor().clamp(0, self.num_train_timesteps - 1).long()
        hi = (lo + 1).clamp(max=self.num_train_timesteps - 1)
        w = (t - lo.double()).clamp(0, 1)
        s = self._sig_all.double()
        return ((1 - w) * s[lo] + w * s[hi]).float()

    def set_timesteps(self, num_inference_steps=None, device=None, timesteps=None):
        T = self.num_train_timesteps
        if timesteps is not None:
            ts = torch.as_tensor(timesteps, dtype=torch.float64).reshape(-1)
        elif self.timestep_spacing == "linspace":
            ts = torch.linspace(0, T - 1, num_inference_steps, dtype=torch.float64).flip(0)
        elif self.timestep_spacing == "leading":
            ts = (torch.arange(0, num_inference_steps, dtype=torch.float64) * (T // num_inference_steps)).round().flip(0) + self.steps_offset
        elif self.timestep_spacing == "trailing":
            ts = torch.arange(T, 0, -T / num_inference_steps, dtype=torch.float64).round() - 1
        else:
            raise ValueError(f"timestep_spacing {self.timestep_spacing!r}")
        self.num_inference_steps = len(ts)
        self.timesteps = ts.float()
        self.sigmas = torch.cat([self._sigma_at(ts), torch.zeros(1)])
        self._step_index = None

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        return m if self.timestep_spacing in ("linspace", "trailing") else (m ** 2 + 1) ** 0.5

    def _index(self, timestep):
        t = float(timestep.reshape(-1)[0]) if isinstance(timestep, torch.Tensor) else float(timestep)
        idx = (self.timesteps - t).abs().argmin().item()
        if abs(float(self.timesteps[idx]) - t) > 1e-3:
            raise ValueError(f"timestep {t} is not on the schedule set by set_timesteps")
        return idx

    def scale_model_input(self, sample, timestep):
        sigma = self.sigmas[self._index(timestep)].to(sample.device, sample.dtype)
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def add_noise(self, original_samples, noise, timesteps):
        sig = torch.stack([self.sigmas[self._index(t)] for t in torch.as_tensor(timesteps).reshape(-1)]).to(original_samples.device, original_samples.dtype)
        while sig.dim() < original_samples.dim():
            sig = sig.unsqueeze(-1)
        return original_samples + noise * sig

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        i = self._index(timestep)
        sigma, sigma_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        if self.prediction_type == "epsilon":
            pred_original = sample - sigma * model_output
        elif self.prediction_type == "v_prediction":
            pred_original = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + sample / (sigma ** 2 + 1)
        else:
            raise ValueError(f"prediction_type {self.prediction_type!r}")
        sigma_up = (sigma_to ** 2 * (sigma ** 2 - sigma_to ** 2) / sigma ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        derivative = (sample - pred_original) / sigma
        noise = torch.randn(model_output.shape, dtype=model_output.dtype, device=model_output.device, generator=generator)
        prev = sample + derivative * (sigma_down - sigma) + noise * sigma_up
        return {'prev_sample': prev, 'pred_original_sample': pred_original}
