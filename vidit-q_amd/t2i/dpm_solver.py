"""DPM-Solver++ (multistep) driver of the t2i sampling loop: the caller of the PixArt hot path.

Restates the ONE configuration the reference's t2i script runs (quant_txt2img.py:130-153):
``DPMS_sigma/DPMS_alpha(model.forward_with_dpmsolver, condition, uncondition, cfg_scale, model_kwargs)
.sample(z, steps, order=2, skip_type="time_uniform", method="multistep")`` = discrete-time VP noise
schedule on the linear betas, noise-prediction model, classifier-free guidance with ONE batched forward
(uncond | cond), data-prediction DPM-Solver++ with ``lower_order_final`` and no final denoise.

References (t2i/diffusion/model/dpm_solver_sigma.py): NoiseScheduleVP :5-170, model_wrapper :172-336,
DPM_Solver.{data_prediction_fn :435, get_time_steps :455, dpm_solver_first_update :551,
multistep_dpm_solver_second_update :805, sample :1069-1262}, interpolate_fn :1288; wrapper
t2i/diffusion/dpm_solver_sigma.py:7-41; beta schedule t2i/diffusion/model/gaussian_diffusion.py
(``get_named_beta_schedule('linear', 1000)``).

Where things run: the schedule scalars are fp32 torch on the host exactly as the reference computes them
(float64 betas -> fp32 ``log_alpha_array``); per step the host passes a handful of floats to elementwise
device ops on the [n, 4, H, W] latent (negligible next to the model forward).  Other solver modes of the
1339-line reference file (singlestep, adaptive, third order, dynamic thresholding, 'taylor') are not part of
this path and raise.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch


def linear_betas(diffusion_steps: int = 1000) -> np.ndarray:
    scale = 1000 / diffusion_steps
    return np.linspace(scale * 0.0001, scale * 0.02, diffusion_steps, dtype=np.float64)


def interpolate_fn(x: torch.Tensor, xp: torch.Tensor, yp: torch.Tensor) -> torch.Tensor:
    """Piecewise-linear f(x) through (xp, yp) (xp ascending, 1-D), linear extrapolation from the outermost
    segment on either side - the value dpm_solver_sigma.py:1288-1327 computes with its sort/gather form."""
    K = xp.numel()
    i = torch.searchsorted(xp, x.reshape(-1), right=False) - 1          # segment [xp[i], xp[i+1]]
    i = i.clamp(0, K - 2)
    x0, x1, y0, y1 = xp[i], xp[i + 1], yp[i], yp[i + 1]
    return (y0 + (x.reshape(-1) - x0) * (y1 - y0) / (x1 - x0)).reshape(x.shape)


class NoiseScheduleVP:
    """schedule='discrete' (dpm_solver_sigma.py:98-106,114-155)."""

    def __init__(self, betas: np.ndarray, dtype=torch.float32):
        b = torch.tensor(betas)                                          # float64, as torch.tensor(np.float64 array)
        log_alphas = 0.5 * torch.log(1 - b).cumsum(dim=0)
        self.T = 1.0
        self.log_alpha_array = self._clip(log_alphas).to(dtype)
        self.total_N = self.log_alpha_array.numel()
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:].to(dtype)

    @staticmethod
    def _clip(log_alphas, clipped_lambda=-5.1):
        log_sigmas = 0.5 * torch.log(1.0 - torch.exp(2.0 * log_alphas))
        lambs = log_alphas - log_sigmas
        idx = int(torch.searchsorted(torch.flip(lambs, [0]), torch.tensor(clipped_lambda, dtype=lambs.dtype)))
        return log_alphas[:-idx] if idx > 0 else log_alphas

    def marginal_log_mean_coeff(self, t):
        return interpolate_fn(t, self.t_array, self.log_alpha_array)

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * torch.log(1.0 - torch.exp(2.0 * lm))


class DPMSolverPP:
    """``DPM_Solver(model_wrapper(...), ns, algorithm_type='dpmsolver++')`` for the configuration above."""

    def __init__(self, model: Callable, condition: torch.Tensor, uncondition: Optional[torch.Tensor], cfg_scale: float,
                 model_kwargs: Optional[dict] = None, diffusion_steps: int = 1000):
        self.model = model
        self.condition, self.uncondition = condition, uncondition
        self.cfg_scale = float(cfg_scale)
        self.model_kwargs = dict(model_kwargs or {})
        self.ns = NoiseScheduleVP(linear_betas(diffusion_steps))
        self._c2 = None                                   # (uncond | cond) text embedding, concatenated once

    # ---- model_wrapper: continuous time -> model input time, classifier-free guidance (:273-332)
    def _noise(self, x, t_cont: torch.Tensor):
        n = x.shape[0]
        # the model-input time is computed on the host in fp32 (the same two roundings as the reference's tensor
        # expression) and materialised by a fill kernel: a `.to(device)` of a host tensor here is a pageable copy that
        # makes the host wait for ALL queued GPU work once per step
        t_val = float((t_cont.float() - 1.0 / self.ns.total_N) * 1000.0)
        if self.cfg_scale == 1.0 or self.uncondition is None:
            return self.model(x, torch.full((n,), t_val, dtype=torch.float32, device=x.device), self.condition,
                              **self.model_kwargs)
        x2 = torch.cat([x, x])
        c, u = self.condition, self.uncondition
        hit = self._c2
        if hit is None or hit[0][0] is not c or hit[0][2] is not u or hit[0][1] != c._version or hit[0][3] != u._version:
            self._c2 = ((c, c._version, u, u._version), torch.cat([u, c]))
        out = self.model(x2, torch.full((2 * n,), t_val, dtype=torch.float32, device=x.device), self._c2[1],
                         **self.model_kwargs)
        noise_uncond, noise = out.chunk(2)
        return noise_uncond + self.cfg_scale * (noise - noise_uncond)

    def _x0(self, x, t):
        """data_prediction_fn (:435-444)."""
        noise = self._noise(x, t)
        alpha_t, sigma_t = float(self.ns.marginal_alpha(t)), float(self.ns.marginal_std(t))
        return (x - sigma_t * noise) / alpha_t

    # ---- updates (host scalars in fp32 as in the reference, applied to device tensors)
    def _first(self, x, s, t, model_s):
        ns = self.ns
        h = ns.marginal_lambda(t) - ns.marginal_lambda(s)
        sigma_s, sigma_t = ns.marginal_std(s), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        phi_1 = torch.expm1(-h)
        return float(sigma_t / sigma_s) * x - float(alpha_t * phi_1) * model_s

    def _second(self, x, model_prev, t_prev, t):
        ns = self.ns
        m1, m0 = model_prev[-2], model_prev[-1]
        t1, t0 = t_prev[-2], t_prev[-1]
        l1, l0, lt = ns.marginal_lambda(t1), ns.marginal_lambda(t0), ns.marginal_lambda(t)
        sigma0, sigma_t = ns.marginal_std(t0), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        h_0, h = l0 - l1, lt - l0
        r0 = h_0 / h
        phi_1 = torch.expm1(-h)
        D1_0 = float(1.0 / r0) * (m0 - m1)
        return float(sigma_t / sigma0) * x - float(alpha_t * phi_1) * m0 - float(0.5 * (alpha_t * phi_1)) * D1_0

    @torch.no_grad()
    def sample(self, x, steps=20, order=2, skip_type="time_uniform", method="multistep", lower_order_final=True,
               step_callback=None):
        if method != "multistep" or skip_type != "time_uniform" or order not in (1, 2):
            raise NotImplementedError("the t2i script runs multistep / time_uniform / order 2 only")
        assert steps >= order
        t_0, t_T = 1.0 / self.ns.total_N, self.ns.T
        ts = torch.linspace(t_T, t_0, steps + 1)                       # fp32, get_time_steps :476
        t = ts[0]
        t_prev, model_prev = [t], [self._x0(x, t)]
        for step in range(1, order):                                   # warm-up with the lower order
            t = ts[step]
            x = self._first(x, t_prev[-1], t, model_prev[-1])
            t_prev.append(t)
            model_prev.append(self._x0(x, t))
        for step in range(order, steps + 1):
            t = ts[step]
            step_order = min(order, steps + 1 - step) if lower_order_final else order
            if step_order == 1:
                x = self._first(x, t_prev[-1], t, model_prev[-1])
            else:
                x = self._second(x, model_prev, t_prev, t)
            for i in range(order - 1):
                t_prev[i], model_prev[i] = t_prev[i + 1], model_prev[i + 1]
            t_prev[-1] = t
            if step < steps:
                model_prev[-1] = self._x0(x, t)
            if step_callback is not None:
                step_callback(step, x)
        return x


def DPMS(model, condition, uncondition, cfg_scale, model_type="noise", noise_schedule="linear",
         guidance_type="classifier-free", model_kwargs=None, diffusion_steps=1000) -> DPMSolverPP:
    """Same call surface as t2i/diffusion/dpm_solver_sigma.py:7-41 (``DPMS_sigma`` / ``DPMS_alpha``)."""
    if model_type != "noise" or noise_schedule != "linear" or guidance_type != "classifier-free":
        raise NotImplementedError("noise-prediction model, linear betas, classifier-free guidance")
    return DPMSolverPP(model, condition, uncondition, cfg_scale, model_kwargs, diffusion_steps)


DPMS_sigma = DPMS
DPMS_alpha = DPMS
