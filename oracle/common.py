"""Shared test fixtures helpers: named configurations, seeded inputs, digests.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from .unet_oracle import UNetCfg

# tiny net used for full-tensor goldens
TINY_CFG = UNetCfg(image_size=8, in_channels=32, out_channels=32, model_channels=32, num_res_blocks=2,
                   channel_mult=(1, 2), attention_resolutions=(1, 2), num_heads=2)
# BASELINE.json configs[0]: 32^3 x 16 plumbing net (apple.yaml net_3d args, resol/feature_size overridden)
PLUMB_CFG = UNetCfg(image_size=32, in_channels=16, out_channels=16, model_channels=64, num_res_blocks=2,
                    channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8), num_heads=2)
# BASELINE.json configs[1]: 64^3 x 32 north star (configs/apple.yaml:228-245)
NORTH_CFG = UNetCfg(image_size=64, in_channels=32, out_channels=32, model_channels=64, num_res_blocks=2,
                    channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8), num_heads=2)


def np_noise(seed: int, shape: Tuple[int, ...]) -> np.ndarray:
    """Standard normal noise from numpy Philox (bit-reproducible across machines)."""
    g = np.random.Generator(np.random.Philox(key=int(seed) & 0xFFFFFFFFFFFFFFFF))
    return g.standard_normal(int(np.prod(shape)), dtype=np.float64).astype(np.float32).reshape(shape)


def seeded_input(cfg: UNetCfg, seed: int, batch: int = 1) -> torch.Tensor:
    r = cfg.image_size
    return torch.from_numpy(np_noise(seed, (batch, cfg.in_channels, r, r, r)))


def digest(y: torch.Tensor) -> Dict[str, np.ndarray]:
    """Few-KB summary of a big (1,C,D,H,W) tensor: per-channel stats, head values, strided probes."""
    c = y.shape[1]
    f = y.reshape(c, -1).double()
    flat = y.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 64).long()
    return {
        "mean": f.mean(1).numpy(), "std": f.std(1).numpy(),
        "min": f.min(1)[0].numpy(), "max": f.max(1)[0].numpy(),
        "head": flat[:4096].numpy().copy(), "probe_idx": idx.numpy(), "probe": flat[idx].numpy().copy(),
    }
