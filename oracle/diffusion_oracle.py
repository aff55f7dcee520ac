"""CPU restatement of the reference DDPM ancestral sampler (START_X / FIXED_SMALL).

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.

Follows (paths relative to /root/reference/holo_diffusion/guided_diffusion):
  * ``gaussian_diffusion.py:25-51``    get_named_beta_schedule ("linear")
  * ``gaussian_diffusion.py:129-187``  GaussianDiffusion.__init__ tables (float64)
  * ``gaussian_diffusion.py:209-227``  q_sample
  * ``gaussian_diffusion.py:229-251``  q_posterior_mean_variance
  * ``gaussian_diffusion.py:253-355``  p_mean_variance (START_X, FIXED_SMALL, clip)
  * ``gaussian_diffusion.py:459-508``  p_sample
  * ``gaussian_diffusion.py:568-643``  p_sample_loop_progressive (incl. max_iter subsampling)
  * ``gaussian_diffusion.py:1046-1059`` _extract_into_tensor (f64 table -> f32 at gather)
"""
from __future__ import annotations

from typing import Callable, Dict, Iterator, List, Optional

import numpy as np
import torch


def linear_betas(num_steps: int, beta_start_unscaled: float = 1e-4, beta_end_unscaled: float = 0.02) -> np.ndarray:
    scale = 1000 / num_steps
    return np.linspace(scale * beta_start_unscaled, scale * beta_end_unscaled, num_steps, dtype=np.float64)


def schedule_tables(betas: np.ndarray) -> Dict[str, np.ndarray]:
    betas = np.array(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    ac_next = np.append(ac[1:], 0.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "alphas_cumprod_next": ac_next,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.append(pv[1], pv[1:])),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }


def _extract(arr: np.ndarray, t: torch.Tensor, shape) -> torch.Tensor:
    res = torch.from_numpy(arr)[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


class DiffusionOracle:
    def __init__(self, num_steps: int = 1000, beta_start_unscaled: float = 1e-4, beta_end_unscaled: float = 0.02):
        self.tables = schedule_tables(linear_betas(num_steps, beta_start_unscaled, beta_end_unscaled))
        self.num_timesteps = num_steps

    def q_sample(self, x_start, t, noise):
        T = self.tables
        return (_extract(T["sqrt_alphas_cumprod"], t, x_start.shape) * x_start
                + _extract(T["sqrt_one_minus_alphas_cumprod"], t, x_start.shape) * noise)

    def p_mean_variance(self, model: Callable, x, t, clip_denoised=True):
        T = self.tables
        out = model(x, t)
        pred_xstart = out.clamp(-1, 1) if clip_denoised else out
        mean = (_extract(T["posterior_mean_coef1"], t, x.shape) * pred_xstart
                + _extract(T["posterior_mean_coef2"], t, x.shape) * x)
        return {
            "mean": mean,
            "variance": _extract(T["posterior_variance"], t, x.shape),
            "log_variance": _extract(T["posterior_log_variance_clipped"], t, x.shape),
            "pred_xstart": pred_xstart,
        }

    def p_sample(self, model, x, t, noise, clip_denoised=True):
        out = self.p_mean_variance(model, x, t, clip_denoised)
        nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        sample = out["mean"] + nonzero * torch.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"], "noise": noise}

    def indices(self, max_iter: Optional[int] = None) -> List[int]:
        idx = list(range(self.num_timesteps))[::-1]
        if max_iter is not None and len(idx) > max_iter:
            if max_iter == 1:
                idx = [idx[0]]
            else:
                idx = [idx[int(i)] for i in torch.round(torch.linspace(0, len(idx) - 1, max_iter)).long()]
        return idx

    @torch.no_grad()
    def p_sample_loop_progressive(self, model, shape, noise_sampler, clip_denoised=True,
                                  max_iter=None) -> Iterator[dict]:
        """noise_sampler(t:int, shape) -> tensor; called with t=num_timesteps for the initial noise."""
        img = noise_sampler(self.num_timesteps, shape)
        for i in self.indices(max_iter):
            t = torch.tensor([i] * shape[0])
            out = self.p_sample(model, img, t, noise_sampler(i, shape), clip_denoised)
            yield out
            img = out["sample"]
