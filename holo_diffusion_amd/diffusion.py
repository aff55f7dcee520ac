"""DDPM ancestral sampler plugin backed by the HIP library.

Mirrors ``ImplicitronGaussianDiffusion`` (/root/reference/holo_diffusion/utils/diffusion_utils.py:89-140),
a Configurable wrapper around ``GaussianDiffusion`` (guided_diffusion/gaussian_diffusion.py):
same config fields (:90-97), same method names and arguments (``q_sample``, ``p_mean_variance``,
``p_sample``, ``p_sample_loop``, ``p_sample_loop_progressive``, ``sample_timesteps``), same
returned dict keys.  Only START_X / FIXED_SMALL (what HoloDiffusion configures, :95-96) is built.

The float64 schedule tables follow gaussian_diffusion.py:25-51,129-187 and are cast to float32 at
gather time like ``_extract_into_tensor`` (:1046-1059).  The per-step elementwise tail
(clamp, posterior mean, noise add; :314-343,237-240,499-506) is one fused HIP kernel
(``holo_ddpm_step``) reading the coefficients by timestep index from a device table, so the
sampling loop never synchronises with the host.
"""
from __future__ import annotations

import enum
import warnings
from typing import Callable, Dict, Iterator, Optional

import numpy as np
import torch

from . import _lib, runtime
from .registry import Configurable, apply_config


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


def get_named_beta_schedule(name: str, num_steps: int, beta_start_unscaled: float, beta_end_unscaled: float) -> np.ndarray:
    if name != "linear":
        raise NotImplementedError(f"unknown/unsupported beta schedule: {name}")
    scale = 1000 / num_steps
    return np.linspace(scale * beta_start_unscaled, scale * beta_end_unscaled, num_steps, dtype=np.float64)


class UniformSampler:
    """guided_diffusion/timestep_sampler.py:67-73 + ScheduleSampler.sample (:40-62)."""

    def __init__(self, num_timesteps: int):
        self._weights = np.ones([num_timesteps])

    def sample(self, batch_size: int, device):
        w = self._weights
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = torch.from_numpy(idx).long().to(device)
        weights = torch.from_numpy(1 / (len(p) * p[idx])).float().to(device)
        return indices, weights


class ImplicitronGaussianDiffusion(Configurable):
    beta_schedule_type: str = "linear"
    num_steps: int = 1000
    beta_start_unscaled: float = 0.0001
    beta_end_unscaled: float = 0.02
    model_mean_type: ModelMeanType = ModelMeanType.START_X
    model_var_type: ModelVarType = ModelVarType.FIXED_SMALL
    schedule_sampler_type: str = "uniform"
    # build-side extension (not a reference field).  None (default): the per-step noise is ``torch.randn_like`` / the
    # caller's ``noise_sampler`` - the reference's draw, the parity path.  An integer: PERF MODE - ``p_sample`` /
    # ``p_sample_loop*`` draw the noise inside the step kernel (``holo_ddpm_step_philox``: Philox4x32-10 keyed on this
    # seed, counter = (element, sample, timestep); no randn launch, the noise never crosses HBM).  Statistically
    # equivalent to, not bit-equal with, torch's generator; the initial x_T still comes from torch.  ``device_noise_stream``
    # separates chains that share a seed (generate.py passes the sample index).
    device_noise_seed: Optional[int] = None
    device_noise_stream: int = 0

    def __init__(self, **kwargs):
        apply_config(self, kwargs)
        if isinstance(self.model_mean_type, str):
            self.model_mean_type = ModelMeanType[self.model_mean_type]
        if isinstance(self.model_var_type, str):
            self.model_var_type = ModelVarType[self.model_var_type]
        if self.model_mean_type != ModelMeanType.START_X or self.model_var_type != ModelVarType.FIXED_SMALL:
            raise NotImplementedError("only model_mean_type=START_X / model_var_type=FIXED_SMALL are on the hot path")
        if self.schedule_sampler_type != "uniform":
            raise NotImplementedError("only the 'uniform' schedule sampler is supported")
        betas = get_named_beta_schedule(self.beta_schedule_type, self.num_steps, self.beta_start_unscaled,
                                        self.beta_end_unscaled)
        self._build_tables(betas)
        self._schedule_sampler = UniformSampler(self.num_timesteps)
        self._dev_tables: Dict[int, torch.Tensor] = {}

    # gaussian_diffusion.py:149-187
    def _build_tables(self, betas: np.ndarray) -> None:
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

    def _tables_on(self, device: torch.device) -> torch.Tensor:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        t = self._dev_tables.get(idx)
        if t is None:
            tab = np.stack([self.posterior_mean_coef1, self.posterior_mean_coef2, self.posterior_log_variance_clipped,
                            np.zeros_like(self.betas)], axis=1).astype(np.float32)
            t = torch.from_numpy(tab).to(device).contiguous()
            self._dev_tables[idx] = t
        return t

    @staticmethod
    def _extract(arr: np.ndarray, timesteps: torch.Tensor, shape) -> torch.Tensor:
        res = torch.from_numpy(arr).to(device=timesteps.device)[timesteps].float()
        while res.dim() < len(shape):
            res = res[..., None]
        return res.expand(shape)

    # ---- forward process (training-side helper; plain torch ops, not on the sampling path) ----
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        return (self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + self._extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    # ---- reverse process -----------------------------------------------------------------------
    def _step(self, x, t, model_output, noise, clip_denoised):
        """Fused HIP tail of p_sample: returns (sample, pred_xstart)."""
        runtime.require_device(x, "ImplicitronGaussianDiffusion")
        L = runtime.lib()
        dev = x.device
        x = x.contiguous()
        model_output = model_output.contiguous()
        noise = noise.contiguous()
        sample = torch.empty_like(x)
        pred = torch.empty_like(x)
        per = x[0].numel()
        _lib.check(L, L.holo_ddpm_step(runtime.ctx(dev), runtime.ptr(self._tables_on(dev)), self.num_timesteps,
                                       runtime.ptr(t), x.shape[0], per, runtime.ptr(x), runtime.ptr(model_output),
                                       runtime.ptr(noise), 1 if clip_denoised else 0, runtime.ptr(sample),
                                       runtime.ptr(pred), runtime.stream_ptr(dev)), "holo_ddpm_step")
        return sample, pred

    def _step_device_noise(self, x, t, model_output, timestep_index: int, clip_denoised, want_pred=True, want_noise=False,
                           channels_last: bool = False):
        """The step with in-kernel Philox noise (perf mode): (sample, pred_xstart | None, noise | None).  A draw is keyed on the
        LOGICAL element (seed, stream, timestep, sample, channel, voxel): ``channels_last`` says which layout ``x`` is in - an
        (N, R, R, R, C) chain and an (N, C, R, R, R) chain of the same seed draw the same noise (C a multiple of 4; otherwise
        the NCDHW tensor's memory order is the key)."""
        runtime.require_device(x, "ImplicitronGaussianDiffusion")
        L = runtime.lib()
        dev = x.device
        x = x.contiguous()
        model_output = model_output.contiguous()
        sample = torch.empty_like(x)
        pred = torch.empty_like(x) if want_pred else None
        noise = torch.empty_like(x) if want_noise else None
        # stream offset: (chain id, timestep) - distinct for every step of every chain that shares the seed
        offset = (int(self.device_noise_stream) << 32) | (int(timestep_index) & 0xFFFFFFFF)
        _lib.check(L, L.holo_ddpm_step_philox(
            runtime.ctx(dev), runtime.ptr(self._tables_on(dev)), self.num_timesteps, runtime.ptr(t), x.shape[0], x[0].numel(),
            runtime.ptr(x), runtime.ptr(model_output), int(self.device_noise_seed) & 0xFFFFFFFFFFFFFFFF, offset,
            1 if clip_denoised else 0, runtime.ptr(sample), runtime.ptr(pred) if want_pred else None,
            runtime.ptr(noise) if want_noise else None,
            0 if (channels_last or x.dim() < 3 or x.shape[1] % 4) else int(x.shape[1]), runtime.stream_ptr(dev)),
            "holo_ddpm_step_philox")
        return sample, pred, noise

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        if model_kwargs is None:
            model_kwargs = {}
        B = x.shape[0]
        assert t.shape == (B,)
        model_output = model(x, t, **model_kwargs)
        if denoised_fn is not None:
            model_output = denoised_fn(model_output)
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        # mean/pred_xstart from the fused kernel with zero noise: sample == mean
        mean, pred = self._step(x, t, model_output, torch.zeros_like(x), clip_denoised)
        return {
            "mean": mean,
            "variance": self._extract(self.posterior_variance, t, x.shape),
            "log_variance": self._extract(self.posterior_log_variance_clipped, t, x.shape),
            "pred_xstart": pred,
        }

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                 noise_sampler=None):
        if cond_fn is not None:
            raise NotImplementedError("cond_fn guidance is not used by HoloDiffusion and is not supported")
        if model_kwargs is None:
            model_kwargs = {}
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        model_output = model(x, t, **model_kwargs)
        if denoised_fn is not None:
            model_output = denoised_fn(model_output)
        if noise_sampler is None and self.device_noise_seed is not None:  # perf mode: noise drawn inside the kernel
            sample, pred, noise = self._step_device_noise(x, t, model_output, int(t[0].item()), clip_denoised)
            return {"sample": sample, "pred_xstart": pred, "noise": noise}  # ("noise": None - it never left the kernel)
        if noise_sampler is not None:
            noise = noise_sampler(int(t[0].item()), x.shape, x.device)  # same host sync as the reference (:495-496)
        else:
            noise = torch.randn_like(x)
        sample, pred = self._step(x, t, model_output, noise, clip_denoised)
        return {"sample": sample, "pred_xstart": pred, "noise": noise}

    def _indices(self, max_iter: Optional[int]):
        indices = list(range(self.num_timesteps))[::-1]
        if max_iter is not None and len(indices) > max_iter:
            warnings.warn(f"Subsampling diffusion steps from {len(indices)} -> {max_iter}")
            if max_iter == 1:
                indices = [indices[0]]
            else:
                indices = [indices[int(i)] for i in torch.round(torch.linspace(0, len(indices) - 1, max_iter)).long()]
        return indices

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, max_iter=None,
                                  noise_sampler=None, _materialize_every_step: bool = True) -> Iterator[dict]:
        """``_materialize_every_step`` (internal, used by ``p_sample_loop``): in the channels-last perf chain (below) yield
        the steps' tensors as NCDHW VIEWS of the channels-last buffers instead of contiguous copies."""
        if cond_fn is not None:
            raise NotImplementedError("cond_fn guidance is not supported")
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        if model_kwargs is None:
            model_kwargs = {}
        if noise is not None:
            img = noise
        elif noise_sampler is not None:
            img = noise_sampler(self.num_timesteps, shape, device)
        else:
            img = torch.randn(*shape, device=device)
        indices = self._indices(max_iter)
        # all timesteps of the chain are uploaded once: no per-step host->device traffic
        ts_all = torch.tensor(indices, dtype=torch.int64, device=device)[:, None].expand(-1, shape[0]).contiguous()
        it = range(len(indices))
        if progress:
            try:
                from tqdm.auto import tqdm
                it = tqdm(it)
            except Exception:
                pass
        # Perf mode (device_noise_seed): the chain stays in the library's channels-last layout - the step kernel is elementwise,
        # hence layout-agnostic, and SimpleUnet3D.forward_channels_last runs without its two layout passes: one conversion
        # at the start of the chain, one per materialised sample (progressive callers get NCDHW-contiguous tensors as ever).
        use_cl = (self.device_noise_seed is not None and noise_sampler is None and denoised_fn is None and not model_kwargs
                  and hasattr(model, "forward_channels_last")
                  and getattr(model, "in_channels", None) == shape[1] and img.is_cuda)
        if use_cl:
            as_ncdhw = (lambda a: a.permute(0, 4, 1, 2, 3).contiguous()) if _materialize_every_step else \
                (lambda a: a.permute(0, 4, 1, 2, 3))
            with torch.no_grad():
                img_cl = img.float().permute(0, 2, 3, 4, 1).contiguous()
                for k in it:
                    t = ts_all[k]
                    out_cl = model.forward_channels_last(img_cl, t)
                    sample_cl, pred_cl, _ = self._step_device_noise(img_cl, t, out_cl, indices[k], clip_denoised,
                                                                    channels_last=True)
                    yield {"sample": as_ncdhw(sample_cl), "pred_xstart": as_ncdhw(pred_cl), "noise": None}
                    img_cl = sample_cl
            return
        with torch.no_grad():
            for k in it:
                t = ts_all[k]
                model_output = model(img, t, **model_kwargs)
                if denoised_fn is not None:
                    model_output = denoised_fn(model_output)
                if noise_sampler is None and self.device_noise_seed is not None:  # perf mode (no host sync: indices[k] is host-side)
                    sample, pred, eps = self._step_device_noise(img, t, model_output, indices[k], clip_denoised)
                elif noise_sampler is not None:
                    eps = noise_sampler(indices[k], img.shape, img.device)
                    sample, pred = self._step(img, t, model_output, eps, clip_denoised)
                else:
                    eps = torch.randn_like(img)
                    sample, pred = self._step(img, t, model_output, eps, clip_denoised)
                yield {"sample": sample, "pred_xstart": pred, "noise": eps}
                img = sample

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, return_all_samples=False, max_iter=None,
                      noise_sampler=None):
        samples = [] if return_all_samples else None
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                     denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                     model_kwargs=model_kwargs, device=device, progress=progress,
                                                     max_iter=max_iter, noise_sampler=noise_sampler,
                                                     _materialize_every_step=return_all_samples):
            if return_all_samples:
                samples.append(sample)
            final = sample["sample"]
        if final is not None and not final.is_contiguous():
            final = final.contiguous()  # (the channels-last perf chain: one conversion at the end of the chain)
        return (final, samples) if return_all_samples else final

    def training_losses(self, *args, **kwargs):
        raise NotImplementedError("training losses are outside the sampling hot path (SURVEY.md §8f)")

    def sample_timesteps(self, *args, **kwargs):
        return self._schedule_sampler.sample(*args, **kwargs)
