"""Per-process device runtime: library handle, context, stream and workspace plumbing.

PyTorch is used here for device memory and streams only (one process per GPU).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib

_CTX: Dict[int, C.c_void_p] = {}


def lib() -> C.CDLL:
    return _lib.load()


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.HoloError(
            f"{what}: tensor is on '{t.device}', but the HoloDiffusion hot path only runs on an MI355X "
            "(HIP) device; there is no CPU fallback.")


def ctx(device: torch.device) -> C.c_void_p:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CTX:
        h = C.c_void_p()
        L = lib()
        _lib.check(L, L.holo_ctx_create(int(idx), C.byref(h)), "holo_ctx_create")
        _CTX[idx] = h
    return _CTX[idx]


def set_deterministic(on: bool, device: torch.device = None) -> bool:
    """Deterministic mode of the scatter-adding backward entries (holo_ctx_set_deterministic, include/holo_abi.h): the grid
    gradient of the renderer's backward and the feature-map gradients of the view-pooling backwards are summed in fixed
    point, bit-identical from run to run.  Returns the previous setting."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    L, h = lib(), ctx(device)
    prev = bool(L.holo_ctx_get_deterministic(h))
    _lib.check(L, L.holo_ctx_set_deterministic(h, 1 if on else 0), "holo_ctx_set_deterministic")
    return prev


@contextlib.contextmanager
def deterministic(on: bool = True, device: torch.device = None):
    """``with runtime.deterministic():`` - the mode above for the calls inside the block."""
    prev = set_deterministic(on, device)
    try:
        yield
    finally:
        set_deterministic(prev, device)


def stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def workspace(owner, device: torch.device, nbytes: int) -> torch.Tensor:
    """A byte buffer of at least ``nbytes`` on ``device``, owned by (and released with) ``owner``."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    held = owner.__dict__.get("_holo_ws")
    if held is None or held[0] != idx or held[1].numel() < nbytes:
        held = (idx, torch.empty(int(nbytes), dtype=torch.uint8, device=device))
        owner.__dict__["_holo_ws"] = held  # bypasses nn.Module.__setattr__: not a buffer, not in the state_dict
    return held[1]


def sync_before_destroy(device) -> None:
    """Kernels still in flight may read the native handle's private weight copies: drain the device first."""
    try:
        if device is not None and torch.cuda.is_available():
            torch.cuda.synchronize(device)
    except Exception:
        pass
