"""DDIM scheduler with the attribute surface the reference's loops touch.

The reference builds `diffusers.DDIMScheduler.from_config(cfg)` with the dict at
`/root/reference/p2p/edit_syn.py:46-57` and then reads (SURVEY.md §8b "Pipeline attrs"):
`set_timesteps`, `timesteps`, `init_noise_sigma`, `step(eps, t, x)["prev_sample"]` /
`.prev_sample` / `['pred_original_sample']`, `config.num_train_timesteps`,
`num_inference_steps`, `alphas_cumprod`, `final_alpha_cumprod`, `scale_model_input`, `order`.

Constants and the update rule restate diffusers' implementation [ext]:
    betas = linspace(sqrt(b0), sqrt(b1), T)^2 ; ac = cumprod(1 - betas)          (fp32)
    timesteps "leading": arange(n) * (T // n), reversed, + steps_offset
    step (eta = 0, epsilon prediction, no clipping):
        x0 = (x - sqrt(1 - a_t) eps) / sqrt(a_t);  x' = sqrt(a_p) x0 + sqrt(1 - a_p) eps
The elementwise update itself runs in the HIP kernel `ief_ddim_step` (csrc/elementwise.hip).
"""
from types import SimpleNamespace

import numpy as np
import torch

from .config import SCHEDULER_CONFIG


class SchedulerOutput(dict):
    """Supports both `out["prev_sample"]` and `out.prev_sample` (both forms occur:
    `/root/reference/p2p/model/sd_utils.py:76`, `/root/reference/p2p/inversion/nti.py:25`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, **config):
        cfg = dict(SCHEDULER_CONFIG)
        cfg.update(config)
        if cfg["beta_schedule"] != "scaled_linear" or cfg["trained_betas"] is not None:
            raise ValueError("only the reference's scaled_linear schedule is supported")
        self.config = SimpleNamespace(**cfg)
        T = cfg["num_train_timesteps"]
        betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, T, dtype=torch.float32) ** 2
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, T)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        return cls(**dict(config))

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts = ts + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    # ------------------------------------------------------------------ coefficients
    def step_coeffs(self, t: int):
        """(a_t, a_prev) as python floats taken from the fp32 table."""
        t = int(t)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p

    def reverse_coeffs(self, t: int):
        """(a_cur, a_next) of `ddim_reverse` (`/root/reference/p2p/inversion/ddim.py:9-18`)."""
        nxt = int(t)
        T = self.config.num_train_timesteps
        cur = min(T - 1, nxt - T // self.num_inference_steps)
        a_c = float(self.alphas_cumprod[cur]) if cur >= 0 else float(self.final_alpha_cumprod)
        return a_c, float(self.alphas_cumprod[nxt])

    @staticmethod
    def linear_form(a_from: float, a_to: float):
        """x' = cx * x + ce * eps  for  x0=(x-sqrt(1-a_from)eps)/sqrt(a_from); x'=sqrt(a_to)x0+sqrt(1-a_to)eps.

        Used by the fused CFG+DDIM kernel, which evaluates x0 and x' in the same operation
        order as the formula above (not this folded form) to stay bit-close to the reference.
        """
        cx = (a_to / a_from) ** 0.5
        ce = (1 - a_to) ** 0.5 - cx * (1 - a_from) ** 0.5
        return cx, ce

    # ------------------------------------------------------------------ update
    def step(self, model_output, timestep, sample, eta: float = 0.0, return_dict: bool = True, **kw):
        if eta != 0.0:
            raise ValueError("only eta = 0 (deterministic DDIM) is on the reference path")
        a_t, a_p = self.step_coeffs(int(timestep))
        if model_output.requires_grad or sample.requires_grad:
            # differentiable form for null-text inversion (`inversion/nti.py:25-28`)
            x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
            prev = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * model_output
        else:
            from . import hip
            prev, x0 = hip.ddim_step(model_output, sample, a_t, a_p)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)
