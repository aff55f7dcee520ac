"""Counter-based synthetic weight generator.

Weights are never shipped: every tensor is regenerated from ``(seed, parameter
name)`` with numpy's Philox bit generator, which is bit-reproducible across
machines.  The same state dict is loaded into the reference modules (only in the
development container, by ``oracle/make_golden.py``), into the CPU oracle and
into the HIP path, so all three see identical parameters.

Scaling follows what the reference does at construction time
(``holo_diffusion/utils/diffusion_utils.py:77-80``: Xavier-uniform on every
Conv3d/Linear, zero biases) except that biases, GroupNorm affine terms and the
zero-initialised attention ``proj_out`` (``unet.py:392``) are randomised as
well, so that parity tests exercise every term of every kernel.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    key = (int(seed) & 0xFFFFFFFF) << 32 | (zlib.crc32(name.encode()) & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def synth_tensor(seed: int, name: str, shape: Tuple[int, ...], kind: str) -> torch.Tensor:
    """kind: 'weight' (Xavier-uniform), 'bias' (small uniform), 'gamma' (1 + small), 'beta' (small)."""
    g = _rng(seed, name)
    n = int(np.prod(shape)) if len(shape) else 1
    u = g.random(n, dtype=np.float64) * 2.0 - 1.0
    if kind == "weight":
        fan_out = shape[0]
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        rf = 1
        if len(shape) > 2:
            rf = int(np.prod(shape[2:]))
            fan_in = shape[1] * rf
            fan_out = shape[0] * rf
        a = np.sqrt(6.0 / (fan_in + fan_out))
        v = u * a
    elif kind == "bias":
        v = u * 0.05
    elif kind == "gamma":
        v = 1.0 + u * 0.2
    elif kind == "beta":
        v = u * 0.1
    else:
        raise ValueError(kind)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def classify(name: str, shape: Tuple[int, ...], norm_names: Iterable[str] = ()) -> str:
    leaf = name.rsplit(".", 1)[-1]
    is_norm = len(shape) == 1 and any(tag in name for tag in (
        ".in_layers.0.", ".out_layers.0.", ".norm.", "out.0."))
    if is_norm:
        return "gamma" if leaf == "weight" else "beta"
    return "weight" if leaf == "weight" else "bias"


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Build a full state dict for the given ``{name: shape}`` map."""
    return {k: synth_tensor(seed, k, tuple(s), classify(k, tuple(s))) for k, s in shapes.items()}
