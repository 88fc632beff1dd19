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
device ops on the [n, 4, H, W] latent (negligible next to the model forward).

Round 6 (review "missing" item 4): the modes of ``DPM_Solver.sample`` (:1069-1279) the t2i script never selects
are here too, for the algorithm the wrappers fix (``algorithm_type="dpmsolver++"``): multistep of order 3
(:864-915), the singlestep solvers of order 1 / 2 / 3 with the reference's order schedule ("DPM-Solver-fast",
:485-543) and its ``singlestep_fixed`` variant, the adaptive step-size solver (:970-1031), ``skip_type``
'logSNR' / 'time_quadratic' (:455-483, ``inverse_lambda`` :157-170), ``solver_type`` 'taylor',
``denoise_to_zero`` (:545-549), ``t_start`` / ``t_end``.  Pinned against the imported reference on an
analytic noise model (tests/golden/dpm_solver_modes.npz, tests/test_oracle_golden_cpu.py).  Still absent:
dynamic thresholding / ``correcting_x0_fn`` / ``correcting_xt_fn`` hooks and the noise-prediction
``algorithm_type="dpmsolver"`` (no wrapper of the reference can select them), SA-Solver.
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

    def inverse_lambda(self, lamb):
        """t of a half-logSNR (discrete schedule, dpm_solver_sigma.py:165-169): log alpha = -softplus(-2 lambda) / 2, then
        the schedule's own table read backwards."""
        lamb = torch.as_tensor(lamb, dtype=self.log_alpha_array.dtype)
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,), dtype=lamb.dtype), -2.0 * lamb)
        return interpolate_fn(log_alpha.reshape(-1), torch.flip(self.log_alpha_array, [0]), torch.flip(self.t_array, [0]))


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

    def _second(self, x, model_prev, t_prev, t, solver_type="dpmsolver"):
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
        if solver_type == "taylor":                                    # (:839-844)
            return float(sigma_t / sigma0) * x - float(alpha_t * phi_1) * m0 + float(alpha_t * (phi_1 / h + 1.0)) * D1_0
        return float(sigma_t / sigma0) * x - float(alpha_t * phi_1) * m0 - float(0.5 * (alpha_t * phi_1)) * D1_0

    def _third(self, x, model_prev, t_prev, t):
        """multistep_dpm_solver_third_update (:864-915), data prediction."""
        ns = self.ns
        m2, m1, m0 = model_prev
        t2, t1, t0 = t_prev
        l2, l1, l0, lt = ns.marginal_lambda(t2), ns.marginal_lambda(t1), ns.marginal_lambda(t0), ns.marginal_lambda(t)
        sigma0, sigma_t = ns.marginal_std(t0), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        h_1, h_0, h = l1 - l2, l0 - l1, lt - l0
        r0, r1 = h_0 / h, h_1 / h
        D1_0 = float(1.0 / r0) * (m0 - m1)
        D1_1 = float(1.0 / r1) * (m1 - m2)
        D1 = D1_0 + float(r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = float(1.0 / (r0 + r1)) * (D1_0 - D1_1)
        phi_1 = torch.expm1(-h)
        phi_2 = phi_1 / h + 1.0
        phi_3 = phi_2 / h - 0.5
        return (float(sigma_t / sigma0) * x - float(alpha_t * phi_1) * m0 + float(alpha_t * phi_2) * D1
                - float(alpha_t * phi_3) * D2)

    def _multistep(self, x, model_prev, t_prev, t, order, solver_type="dpmsolver"):
        if order == 1:
            return self._first(x, t_prev[-1], t, model_prev[-1])
        if order == 2:
            return self._second(x, model_prev, t_prev, t, solver_type)
        return self._third(x, model_prev, t_prev, t)

    # ---- singlestep solvers (:598-803): intermediate times on the half-logSNR axis, one model call per stage
    def _single2(self, x, s, t, r1=None, model_s=None, solver_type="dpmsolver", want=False):
        ns = self.ns
        r1 = 0.5 if r1 is None else r1
        ls, lt = ns.marginal_lambda(s), ns.marginal_lambda(t)
        h = lt - ls
        s1 = ns.inverse_lambda(ls + r1 * h).reshape(())
        sigma_s, sigma_s1, sigma_t = ns.marginal_std(s), ns.marginal_std(s1), ns.marginal_std(t)
        alpha_s1, alpha_t = torch.exp(ns.marginal_log_mean_coeff(s1)), torch.exp(ns.marginal_log_mean_coeff(t))
        phi_11, phi_1 = torch.expm1(-r1 * h), torch.expm1(-h)
        if model_s is None:
            model_s = self._x0(x, s)
        x_s1 = float(sigma_s1 / sigma_s) * x - float(alpha_s1 * phi_11) * model_s
        model_s1 = self._x0(x_s1, s1)
        if solver_type == "dpmsolver":
            x_t = (float(sigma_t / sigma_s) * x - float(alpha_t * phi_1) * model_s
                   - float((0.5 / r1) * (alpha_t * phi_1)) * (model_s1 - model_s))
        else:                                                          # 'taylor'
            x_t = (float(sigma_t / sigma_s) * x - float(alpha_t * phi_1) * model_s
                   + float((1.0 / r1) * (alpha_t * (phi_1 / h + 1.0))) * (model_s1 - model_s))
        return (x_t, dict(model_s=model_s, model_s1=model_s1)) if want else x_t

    def _single3(self, x, s, t, r1=None, r2=None, model_s=None, model_s1=None, solver_type="dpmsolver"):
        ns = self.ns
        r1 = 1.0 / 3.0 if r1 is None else r1
        r2 = 2.0 / 3.0 if r2 is None else r2
        ls, lt = ns.marginal_lambda(s), ns.marginal_lambda(t)
        h = lt - ls
        s1, s2 = ns.inverse_lambda(ls + r1 * h).reshape(()), ns.inverse_lambda(ls + r2 * h).reshape(())
        sigma_s, sigma_s1, sigma_s2, sigma_t = ns.marginal_std(s), ns.marginal_std(s1), ns.marginal_std(s2), ns.marginal_std(t)
        alpha_s1, alpha_s2, alpha_t = (torch.exp(ns.marginal_log_mean_coeff(s1)), torch.exp(ns.marginal_log_mean_coeff(s2)),
                                       torch.exp(ns.marginal_log_mean_coeff(t)))
        phi_11, phi_12, phi_1 = torch.expm1(-r1 * h), torch.expm1(-r2 * h), torch.expm1(-h)
        phi_22 = torch.expm1(-r2 * h) / (r2 * h) + 1.0
        phi_2 = phi_1 / h + 1.0
        phi_3 = phi_2 / h - 0.5
        if model_s is None:
            model_s = self._x0(x, s)
        if model_s1 is None:
            x_s1 = float(sigma_s1 / sigma_s) * x - float(alpha_s1 * phi_11) * model_s
            model_s1 = self._x0(x_s1, s1)
        x_s2 = (float(sigma_s2 / sigma_s) * x - float(alpha_s2 * phi_12) * model_s
                + float(r2 / r1 * (alpha_s2 * phi_22)) * (model_s1 - model_s))
        model_s2 = self._x0(x_s2, s2)
        if solver_type == "dpmsolver":
            return (float(sigma_t / sigma_s) * x - float(alpha_t * phi_1) * model_s
                    + float((1.0 / r2) * (alpha_t * phi_2)) * (model_s2 - model_s))
        D1_0 = float(1.0 / r1) * (model_s1 - model_s)                  # 'taylor'
        D1_1 = float(1.0 / r2) * (model_s2 - model_s)
        D1 = (float(r2) * D1_0 - float(r1) * D1_1) / float(r2 - r1)
        D2 = 2.0 * (D1_1 - D1_0) / float(r2 - r1)
        return (float(sigma_t / sigma_s) * x - float(alpha_t * phi_1) * model_s + float(alpha_t * phi_2) * D1
                - float(alpha_t * phi_3) * D2)

    def _singlestep(self, x, s, t, order, solver_type, r1, r2):
        if order == 1:
            return self._first(x, s, t, self._x0(x, s))
        if order == 2:
            return self._single2(x, s, t, r1=r1, solver_type=solver_type)
        return self._single3(x, s, t, r1=r1, r2=r2, solver_type=solver_type)

    def time_steps(self, skip_type, t_T, t_0, N):
        """get_time_steps (:455-483): N + 1 fp32 times from t_T down to t_0."""
        if skip_type == "logSNR":
            lam_T, lam_0 = self.ns.marginal_lambda(torch.tensor(t_T)), self.ns.marginal_lambda(torch.tensor(t_0))
            return self.ns.inverse_lambda(torch.linspace(float(lam_T), float(lam_0), N + 1))
        if skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        if skip_type == "time_quadratic":
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
        raise ValueError("skip_type is 'logSNR', 'time_uniform' or 'time_quadratic', got %r" % (skip_type,))

    def singlestep_schedule(self, steps, order, skip_type, t_T, t_0):
        """Orders and outer times that spend exactly `steps` model calls (:485-543)."""
        if order == 3:
            K = steps // 3 + 1
            orders = [3] * (K - 2) + [2, 1] if steps % 3 == 0 else [3] * (K - 1) + ([1] if steps % 3 == 1 else [2])
        elif order == 2:
            K = steps // 2 + steps % 2
            orders = [2] * (steps // 2) + [1] * (steps % 2)
        elif order == 1:
            K, orders = 1, [1] * steps
        else:
            raise ValueError("'order' must be 1, 2 or 3")
        if skip_type == "logSNR":                                      # (the reference's K here, K = 1 for order 1 included)
            return self.time_steps(skip_type, t_T, t_0, K), orders
        idx = torch.cumsum(torch.tensor([0] + orders), 0)
        return self.time_steps(skip_type, t_T, t_0, steps)[idx], orders

    def _adaptive(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5, solver_type="dpmsolver"):
        """dpm_solver_adaptive (:970-1031): a lower / higher order pair per trial step, accepted when the scaled difference
        is <= 1; the step on the half-logSNR axis follows the error estimate.  (One host read of E per trial step.)"""
        ns = self.ns
        if order not in (2, 3):
            raise ValueError("the adaptive solver has order 2 or 3, got %r" % (order,))
        s = torch.tensor(t_T, dtype=torch.float32)
        lam_s, lam_0 = ns.marginal_lambda(s), ns.marginal_lambda(torch.tensor(t_0, dtype=torch.float32))
        h = torch.tensor(h_init, dtype=torch.float32)
        x_prev = x
        self.nfe = 0
        while float(torch.abs(s - t_0)) > t_err:
            t = ns.inverse_lambda(lam_s + h).reshape(())
            if order == 2:
                model_s = self._x0(x, s)
                x_lower = self._first(x, s, t, model_s)
                x_higher = self._single2(x, s, t, r1=0.5, model_s=model_s, solver_type=solver_type)
            else:
                x_lower, kw = self._single2(x, s, t, r1=1.0 / 3.0, solver_type=solver_type, want=True)
                x_higher = self._single3(x, s, t, r1=1.0 / 3.0, r2=2.0 / 3.0, solver_type=solver_type, **kw)
            delta = torch.max(torch.ones_like(x) * atol, rtol * torch.max(torch.abs(x_lower), torch.abs(x_prev)))
            v = (x_higher - x_lower) / delta
            E = torch.sqrt(torch.square(v.reshape(v.shape[0], -1)).mean(dim=-1, keepdim=True)).max().float().cpu()
            if bool(E <= 1.0):
                x, s, x_prev = x_higher, t, x_lower
                lam_s = ns.marginal_lambda(s)
            h = torch.min(theta * h * torch.float_power(E, -1.0 / order).float(), lam_0 - lam_s)
            self.nfe += order
        return x

    @torch.no_grad()
    def sample(self, x, steps=20, t_start=None, t_end=None, order=2, skip_type="time_uniform", method="multistep",
               lower_order_final=True, denoise_to_zero=False, solver_type="dpmsolver", atol=0.0078, rtol=0.05,
               step_callback=None):
        """``DPM_Solver.sample`` (:1069-1279) for data-prediction DPM-Solver++.  ``method``: 'multistep' (order 1-3),
        'singlestep', 'singlestep_fixed', 'adaptive' (order 2 / 3)."""
        if solver_type not in ("dpmsolver", "taylor"):
            raise ValueError("'solver_type' is 'dpmsolver' or 'taylor', got %r" % (solver_type,))
        if order not in (1, 2, 3):
            raise ValueError("'order' must be 1, 2 or 3")
        t_0 = 1.0 / self.ns.total_N if t_end is None else t_end
        t_T = self.ns.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0
        if method == "adaptive":
            x = self._adaptive(x, order, t_T, t_0, atol=atol, rtol=rtol, solver_type=solver_type)
        elif method == "multistep":
            assert steps >= order
            ts = self.time_steps(skip_type, t_T, t_0, steps)           # fp32, get_time_steps :455
            t = ts[0]
            t_prev, model_prev = [t], [self._x0(x, t)]
            for step in range(1, order):                               # warm-up with the lower orders
                t = ts[step]
                x = self._multistep(x, model_prev, t_prev, t, step, solver_type)
                t_prev.append(t)
                model_prev.append(self._x0(x, t))
            for step in range(order, steps + 1):
                t = ts[step]
                step_order = min(order, steps + 1 - step) if lower_order_final else order
                x = self._multistep(x, model_prev[-step_order:] if step_order > 1 else model_prev,
                                    t_prev[-step_order:] if step_order > 1 else t_prev, t, step_order, solver_type)
                for i in range(order - 1):
                    t_prev[i], model_prev[i] = t_prev[i + 1], model_prev[i + 1]
                t_prev[-1] = t
                if step < steps:
                    model_prev[-1] = self._x0(x, t)
                if step_callback is not None:
                    step_callback(step, x)
        elif method in ("singlestep", "singlestep_fixed"):
            if method == "singlestep":
                outer, orders = self.singlestep_schedule(steps, order, skip_type, t_T, t_0)
            else:
                K = steps // order
                orders, outer = [order] * K, self.time_steps(skip_type, t_T, t_0, K)
            for step, o in enumerate(orders):
                s_, t_ = outer[step], outer[step + 1]
                lam = self.ns.marginal_lambda(self.time_steps(skip_type, float(s_), float(t_), o))
                h = lam[-1] - lam[0]
                r1 = None if o <= 1 else (lam[1] - lam[0]) / h
                r2 = None if o <= 2 else (lam[2] - lam[0]) / h
                x = self._singlestep(x, s_, t_, o, solver_type, r1, r2)
                if step_callback is not None:
                    step_callback(step + 1, x)
        else:
            raise ValueError("method is 'multistep', 'singlestep', 'singlestep_fixed' or 'adaptive', got %r" % (method,))
        if denoise_to_zero:
            x = self._x0(x, torch.tensor(t_0, dtype=torch.float32))
        return x


def DPMS(model, condition, uncondition, cfg_scale, model_type="noise", noise_schedule="linear",
         guidance_type="classifier-free", model_kwargs=None, diffusion_steps=1000) -> DPMSolverPP:
    """Same call surface as t2i/diffusion/dpm_solver_sigma.py:7-41 (``DPMS_sigma`` / ``DPMS_alpha``)."""
    if model_type != "noise" or noise_schedule != "linear" or guidance_type != "classifier-free":
        raise NotImplementedError("noise-prediction model, linear betas, classifier-free guidance")
    return DPMSolverPP(model, condition, uncondition, cfg_scale, model_kwargs, diffusion_steps)


DPMS_sigma = DPMS
DPMS_alpha = DPMS
