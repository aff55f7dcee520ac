"""Denoiser plugin: ``SimpleUnet3D`` backed by the HIP library.

Mirrors /root/reference/holo_diffusion/utils/diffusion_utils.py:30-86:
  * ``Unet3DBase(ReplaceableBase, torch.nn.Module)`` with
    ``forward(x, timesteps, cond_features=None, **kwargs)`` (:30-38)
  * ``@registry.register class SimpleUnet3D(Unet3DBase)`` with the config fields of :43-53 and
    ``forward`` = optional channel concat of ``cond_features`` then ``self._net(x, timesteps)`` (:82-86)

The parameters live in ordinary ``torch.nn.Parameter`` tensors under ``_net.<guided-diffusion name>``
so that ``state_dict()`` / ``load_state_dict()`` are interchangeable with the reference's checkpoints
(``net_3d._net.*`` keys, trainer/model_factory.py:115-126).  The arithmetic of ``UNetModel.forward``
(guided_diffusion/unet.py:800-837) runs entirely in ``libholo_mi355x.so`` (``holo_unet_forward``).
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, Optional, Tuple

import torch

from . import _lib, runtime
from .registry import ReplaceableBase, apply_config, registry


class Unet3DBase(ReplaceableBase, torch.nn.Module):
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, cond_features: Optional[torch.Tensor] = None,
                **kwargs) -> torch.Tensor:
        raise NotImplementedError()


class _Node(torch.nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _add_param(root: torch.nn.Module, dotted: str, p: torch.nn.Parameter) -> None:
    parts = dotted.split(".")
    m = root
    for part in parts[:-1]:
        if part not in m._modules:
            m.add_module(part, _Node())
        m = m._modules[part]
    m.register_parameter(parts[-1], p)


def _init_param(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """Initialisation as the reference leaves it after SimpleUnet3D.__post_init__
    (diffusion_utils.py:77-80: Xavier-uniform Conv3d/Linear weights, zero biases; GroupNorm 1/0;
    Conv1d qkv keeps torch's default init; proj_out is zero, unet.py:392)."""
    leaf = name.rsplit(".", 1)[-1]
    is_norm = len(shape) == 1 and any(s in name for s in (".in_layers.0.", ".out_layers.0.", ".norm.", "out.0."))
    if is_norm:
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    if ".proj_out." in name:
        return torch.zeros(shape)
    if ".qkv." in name:
        fan_in = shape[1] if len(shape) > 1 else shape[0]
        bound = 1.0 / math.sqrt(fan_in)
        return torch.empty(shape).uniform_(-bound, bound)
    if leaf == "bias":
        return torch.zeros(shape)
    w = torch.empty(shape)
    torch.nn.init.xavier_uniform_(w)
    return w


@registry.register
class SimpleUnet3D(Unet3DBase):
    image_size: int = 64
    in_channels: int = 128
    out_channels: int = 128
    model_channels: int = 128
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 8)
    attention_resolutions: Tuple[int, ...] = (8, 16)
    num_heads: int = 2
    dropout: float = 0.0
    # 3d down/upsamples have the same size in all 3 dims
    homogeneous_resample: bool = True
    # build-side extension (not a reference field): arithmetic of the stride-1 3x3x3 convolutions, "f32" (exact fp32
    # MFMA, the reference's arithmetic) or "bf16" (activations stored as bf16 in HBM, bf16 products on the matrix
    # cores with fp32 accumulation, fp32 GroupNorm statistics, fp32 network input / output; opt-in for
    # the bf16 configurations, tolerance rtol 2e-2) or "f32_bf16x3" (fp32 operands split exactly into three bf16
    # terms, six bf16 MFMAs per product: fp32-accurate on the bf16 matrix cores) - holo_unet_set_compute_dtype
    compute_dtype: str = "f32"
    # build-side extension: a differentiable call (grad mode on, a parameter or the input requires grad) runs the TAPED forward,
    # which allocates the training workspace (every intermediate kept; 64^3 x 32: ~6 GB) and prepares the transposed
    # convolution weights, so that loss.backward() needs no second forward.  False: such a call costs the inference
    # workspace only and its backward re-runs the forward (holo_unet_backward).  (HOLO_NO_AUTOGRAD_TAPE=1: same, by env.)
    autograd_tape: bool = True

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        apply_config(self, kwargs)
        if self.dropout != 0.0:
            raise _lib.HoloError("SimpleUnet3D: dropout must be 0 on the sampling path")
        self._net = _Node()
        self._handle: Optional[C.c_void_p] = None
        self._handle_device: Optional[torch.device] = None
        self._dirty = True
        self._param_names = []
        self._create_parameters()

    # ---- parameter tree -------------------------------------------------------------------
    def _cfg_struct(self, size: Optional[int] = None):
        return _lib.make_unet_cfg(size or self.image_size, self.in_channels, self.out_channels, self.model_channels,
                                  self.num_res_blocks, self.channel_mult, self.attention_resolutions, self.num_heads,
                                  self.homogeneous_resample)

    def _create_parameters(self) -> None:
        """Parameter names/shapes come from the library's own enumeration (single source of truth)."""
        for name, shape in self.param_shapes().items():
            _add_param(self._net, name, torch.nn.Parameter(_init_param(name, shape), requires_grad=False))
            self._param_names.append(name)

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        from .structure import unet_param_shapes
        return unet_param_shapes(self.image_size, self.in_channels, self.out_channels, self.model_channels,
                                 self.num_res_blocks, self.channel_mult, self.attention_resolutions)

    def _mark_dirty(self) -> None:
        self._dirty = True
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1  # read by HoloDiffusionModel's refine cache

    def _apply(self, fn, *a, **k):  # .to()/.cuda()/.float() move the tensors: re-bind on next forward
        self._mark_dirty()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._mark_dirty()
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        # a PARENT module's load_state_dict (e.g. HoloDiffusionModel) recurses through _load_from_state_dict of every
        # sub-module and never calls the child's load_state_dict: this is the hook that sees those loads
        self._mark_dirty()
        return super()._load_from_state_dict(*a, **k)

    def _poll_parameter_versions(self, sd=None):
        """In-place parameter updates (``opt.step()``, ``p.add_()``) go through none of the hooks above: every tensor's
        (storage, version counter) is compared with what the library's packed copies were made from; a difference marks
        the copies dirty and advances ``_weights_epoch``.  Returns the current fingerprint."""
        if sd is None:
            sd = dict(self._net.named_parameters())
        versions = tuple((sd[k].data_ptr(), sd[k]._version) for k in self._param_names)
        if not self._dirty and versions != self.__dict__.get("_param_versions"):
            self._mark_dirty()
        return versions

    def weights_epoch(self) -> int:
        """Counter that advances whenever the parameters changed by any route (the key of caches of this net's outputs)."""
        self._poll_parameter_versions()
        return getattr(self, "_weights_epoch", 0)

    def mark_parameters_changed(self) -> None:
        """Forces a re-pack of the library's private copies.  Not needed after ordinary in-place updates (``opt.step()``,
        ``p.add_()``): ``_ensure_handle`` compares every parameter's storage and version counter on each call; this is for
        writes that bypass the version counter (``p.data`` views, raw pointers)."""
        self._mark_dirty()

    # ---- native handle --------------------------------------------------------------------
    def _ensure_handle(self, device: torch.device, size: Optional[int] = None) -> C.c_void_p:
        """The native handle for inputs of ``size``^3 (default: ``image_size``).  The reference's UNetModel is fully
        convolutional - ``image_size`` is a stored field, its own test feeds 32^3 grids to a net built with the default
        64 (tests/test_diffusion_utils.py:16-30) - while the native plan is built for one grid size; a forward at
        another size switches the handle to a plan of that size (weights are re-bound once per switch)."""
        L = runtime.lib()
        size = int(size or getattr(self, "_handle_size", None) or self.image_size)  # no size: keep the live plan
        if self._handle is None or self._handle_device != device or getattr(self, "_handle_size", None) != size:
            if self._handle is not None:
                runtime.sync_before_destroy(self._handle_device)
                L.holo_unet_destroy(self._handle)
                self._handle = None
            # a tape of the old handle (forward_train) dies with it: backward_taped must not see the freed pointer
            self.__dict__.pop("_holo_tape", None)
            h = C.c_void_p()
            cfg = self._cfg_struct(size)
            _lib.check(L, L.holo_unet_create(runtime.ctx(device), C.byref(cfg), C.byref(h)), "holo_unet_create")
            # cross-check the library's enumeration against the Python tree
            n = L.holo_unet_num_params(h)
            name = C.create_string_buffer(256)
            shp = (C.c_int64 * 8)()
            nd = C.c_int()
            shapes = self.param_shapes()
            if n != len(shapes):
                raise _lib.HoloError(f"parameter count mismatch: library {n}, python {len(shapes)}")
            for i in range(n):
                _lib.check(L, L.holo_unet_param_info(h, i, name, 256, shp, C.byref(nd)), "holo_unet_param_info")
                k = name.value.decode()
                if tuple(shp[:nd.value]) != tuple(shapes.get(k, ())):
                    raise _lib.HoloError(f"parameter '{k}': library shape {tuple(shp[:nd.value])} != {shapes.get(k)}")
            self._handle, self._handle_device, self._handle_size, self._dirty = h, device, size, True
            # a new native handle holds no transposed (dgrad) weights, whatever address malloc gave it
            self.__dict__.pop("_dgrad_key", None)
            self.__dict__["_handle_generation"] = self.__dict__.get("_handle_generation", 0) + 1
        code = {"f32": _lib.HOLO_DTYPE_F32, "bf16": _lib.HOLO_DTYPE_BF16,
                "f32_bf16x3": _lib.HOLO_DTYPE_F32_BF16X3}.get(self.compute_dtype)
        if code is None:
            raise _lib.HoloError("SimpleUnet3D.compute_dtype must be 'f32', 'bf16' or 'f32_bf16x3' "
                                 f"(got {self.compute_dtype!r})")
        _lib.check(L, L.holo_unet_set_compute_dtype(self._handle, code), "holo_unet_set_compute_dtype")
        sd = dict(self._net.named_parameters())
        versions = self._poll_parameter_versions(sd)
        if self._dirty:
            # re-packing the forward weights under a live tape would pair new weights with the old taped activations (and
            # stale dgrad weights) in backward_taped: the tape is dropped, the autograd node then re-runs its forward
            self.__dict__.pop("_holo_tape", None)
            st = runtime.stream_ptr(device)
            for k in self._param_names:
                p = sd[k]
                if p.device != device or p.dtype != torch.float32:
                    raise _lib.HoloError(f"parameter '{k}' is {p.dtype} on {p.device}; expected float32 on {device}")
                t = p.detach().contiguous()
                _lib.check(L, L.holo_unet_set_param(self._handle, k.encode(), runtime.ptr(t), _lib.HOLO_DTYPE_F32,
                                                   t.dim(), _lib.shape_array(t.shape), st), f"holo_unet_set_param({k})")
            torch.cuda.current_stream(device).synchronize()  # temporaries from .contiguous() must outlive the copies
            self._dirty = False
            self.__dict__["_param_versions"] = versions
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                runtime.sync_before_destroy(self._handle_device)
                runtime.lib().holo_unet_destroy(self._handle)
        except Exception:
            pass

    # ---- forward --------------------------------------------------------------------------
    def forward(self, x, timesteps, cond_features=None, **kwargs):
        if cond_features is not None:
            x = torch.cat([x, cond_features], dim=1)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._net.parameters())):
            # differentiable call: `loss.backward()` reaches holo_unet_backward through the autograd node below
            names = list(self._param_names)
            sd = dict(self._net.named_parameters())
            return _HoloUnetFn.apply(self, x, timesteps, names, *[sd[k] for k in names])
        return self._forward_impl(x, timesteps)

    def _forward_impl(self, x, timesteps):
        runtime.require_device(x, "SimpleUnet3D.forward")
        if x.dim() != 5 or x.shape[1] != self.in_channels or len(set(x.shape[2:])) != 1 or \
                x.shape[2] % (1 << (len(self.channel_mult) - 1)):
            raise _lib.HoloError(f"SimpleUnet3D.forward: expected (N,{self.in_channels},R,R,R) with R a multiple of "
                                 f"{1 << (len(self.channel_mult) - 1)}, got {tuple(x.shape)}")
        dev = x.device
        h = self._ensure_handle(dev, int(x.shape[2]))
        L = runtime.lib()
        x = x.contiguous().float()
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        B = x.shape[0]
        if t.shape != (B,):
            raise _lib.HoloError("SimpleUnet3D.forward: timesteps must have shape (N,)")
        nbytes = L.holo_unet_workspace_bytes(h, B)
        ws = runtime.workspace(self, dev, nbytes)
        y = torch.empty((B, self.out_channels) + tuple(x.shape[2:]), dtype=torch.float32, device=dev)
        _lib.check(L, L.holo_unet_forward(h, B, runtime.ptr(x), runtime.ptr(t), runtime.ptr(y), runtime.ptr(ws),
                                          ws.numel(), runtime.stream_ptr(dev)), "holo_unet_forward")
        return y

    @torch.no_grad()
    def forward_channels_last(self, x_cl: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """``forward`` on channels-last tensors (``holo_unet_forward_cl``): ``x_cl`` is (N, R, R, R, C) contiguous fp32 - the
        library's own layout -, so is the result; the two layout passes of a plain call do not run (in the bf16 storage mode an
        element-wise fp32 -> bf16 cast takes the place of the first).  The sampler's perf mode keeps its chain in this form
        (``ImplicitronGaussianDiffusion.p_sample_loop`` with ``device_noise_seed``)."""
        runtime.require_device(x_cl, "SimpleUnet3D.forward_channels_last")
        if x_cl.dim() != 5 or x_cl.shape[4] != self.in_channels or len(set(x_cl.shape[1:4])) != 1 or \
                x_cl.shape[1] % (1 << (len(self.channel_mult) - 1)) or not x_cl.is_contiguous() or x_cl.dtype != torch.float32:
            raise _lib.HoloError(f"SimpleUnet3D.forward_channels_last: expected a contiguous float32 (N,R,R,R,{self.in_channels}), "
                                 f"got {tuple(x_cl.shape)} {x_cl.dtype}")
        dev = x_cl.device
        h = self._ensure_handle(dev, int(x_cl.shape[1]))
        L = runtime.lib()
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        B = x_cl.shape[0]
        if t.shape != (B,):
            raise _lib.HoloError("SimpleUnet3D.forward_channels_last: timesteps must have shape (N,)")
        ws = runtime.workspace(self, dev, L.holo_unet_workspace_bytes(h, B))
        y = torch.empty(tuple(x_cl.shape[:4]) + (self.out_channels,), dtype=torch.float32, device=dev)
        _lib.check(L, L.holo_unet_forward_cl(h, B, runtime.ptr(x_cl), runtime.ptr(t), runtime.ptr(y), runtime.ptr(ws),
                                             ws.numel(), runtime.stream_ptr(dev)), "holo_unet_forward_cl")
        return y

    # ---- backward -------------------------------------------------------------
    def _ensure_dgrad_weights(self, device: torch.device) -> None:
        """Weights of the transposed convolutions (holo_unet_set_dgrad_weight), re-prepared when a parameter changed."""
        h = self._ensure_handle(device)
        key = tuple((k, p.data_ptr(), p._version) for k, p in self._net.named_parameters() if p.dim() >= 3)
        gen = self.__dict__.get("_handle_generation", 0)
        if self.__dict__.get("_dgrad_key") == (gen, key):
            return
        L = runtime.lib()
        st = runtime.stream_ptr(device)
        keep = []
        for k, p in self._net.named_parameters():
            if p.dim() >= 3:
                t = p.detach().contiguous()
                keep.append(t)
                _lib.check(L, L.holo_unet_set_dgrad_weight(h, k.encode(), runtime.ptr(t), st), f"holo_unet_set_dgrad_weight({k})")
        torch.cuda.current_stream(device).synchronize()
        self.__dict__["_dgrad_key"] = (gen, key)

    @torch.no_grad()
    def backward(self, x: torch.Tensor, timesteps: torch.Tensor, grad_output: torch.Tensor, params=None):
        """Gradients of ``(forward(x, timesteps) * grad_output).sum()``: returns ``(y, grad_x, {parameter name: gradient})``
        with the gradients in the reference's parameter layouts (``_net.<guided-diffusion name>`` without the prefix).
        ``params``: names to fetch (default: all).  The forward is re-run inside the call with every intermediate kept
        (``holo_unet_backward``); fp32 mode only."""
        runtime.require_device(x, "SimpleUnet3D.backward")
        if self.compute_dtype != "f32":
            raise _lib.HoloError("SimpleUnet3D.backward runs in the fp32 mode only")
        dev = x.device
        h = self._ensure_handle(dev, int(x.shape[2]))
        self._ensure_dgrad_weights(dev)
        L = runtime.lib()
        x = x.contiguous().float()
        g = grad_output.contiguous().float()
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        B = x.shape[0]
        if tuple(g.shape) != (B, self.out_channels) + tuple(x.shape[2:]):
            raise _lib.HoloError(f"grad_output must be {(B, self.out_channels) + tuple(x.shape[2:])}, got {tuple(g.shape)}")
        held = self._train_ws(h, B, dev)
        self.__dict__.pop("_holo_tape", None)  # (this call re-tapes the workspace)
        y = torch.empty((B, self.out_channels) + tuple(x.shape[2:]), device=dev)
        gx = torch.empty_like(x)
        st = runtime.stream_ptr(dev)
        _lib.check(L, L.holo_unet_backward(h, B, runtime.ptr(x), runtime.ptr(t), runtime.ptr(g), runtime.ptr(y), runtime.ptr(gx),
                                           runtime.ptr(held), held.numel(), st), "holo_unet_backward")
        grads = {}
        shapes = self.param_shapes()
        for k in (params if params is not None else self._param_names):
            out = torch.empty(shapes[k], device=dev)
            _lib.check(L, L.holo_unet_get_grad(h, k.encode(), runtime.ptr(out), out.numel(), runtime.ptr(held), st),
                       f"holo_unet_get_grad({k})")
            grads[k] = out
        return y, gx, grads

    def _train_ws(self, h, B: int, dev):
        L = runtime.lib()
        nbytes = L.holo_unet_backward_workspace_bytes(h, B)
        if nbytes == 0:
            raise _lib.HoloError("holo_unet_backward_workspace_bytes: " + (L.holo_last_error() or b"").decode())
        held = self.__dict__.get("_holo_train_ws")
        if held is None or held.device != dev or held.numel() < nbytes:
            held = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            self.__dict__["_holo_train_ws"] = held
        return held

    @torch.no_grad()
    def forward_train(self, x: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """The taped forward (``holo_unet_forward_train``): returns y and leaves every intermediate in the training workspace
        for ONE following ``backward_taped`` - for a cotangent that depends on y (the clamp in ``training_backward``), where
        ``backward`` would pay a second forward.  fp32 mode only."""
        runtime.require_device(x, "SimpleUnet3D.forward_train")
        if self.compute_dtype != "f32":
            raise _lib.HoloError("SimpleUnet3D.forward_train runs in the fp32 mode only")
        if x.dim() != 5 or x.shape[1] != self.in_channels or len(set(x.shape[2:])) != 1 or \
                x.shape[2] % (1 << (len(self.channel_mult) - 1)) or tuple(timesteps.shape) != (x.shape[0],):
            raise _lib.HoloError(f"SimpleUnet3D.forward_train: expected (N,{self.in_channels},R,R,R) with R a multiple of "
                                 f"{1 << (len(self.channel_mult) - 1)} and timesteps (N,), got {tuple(x.shape)}, {tuple(timesteps.shape)}")
        dev = x.device
        h = self._ensure_handle(dev, int(x.shape[2]))
        self._ensure_dgrad_weights(dev)
        L = runtime.lib()
        x = x.contiguous().float()
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        B = x.shape[0]
        held = self._train_ws(h, B, dev)
        y = torch.empty((B, self.out_channels) + tuple(x.shape[2:]), device=dev)
        _lib.check(L, L.holo_unet_forward_train(h, B, runtime.ptr(x), runtime.ptr(t), runtime.ptr(y), runtime.ptr(held),
                                                held.numel(), runtime.stream_ptr(dev)), "holo_unet_forward_train")
        self.__dict__["_holo_tape"] = (x, t, h)  # (kept alive: the backward reads the timesteps again)
        self.__dict__["_holo_tape_id"] = self.__dict__.get("_holo_tape_id", 0) + 1
        return y

    @torch.no_grad()
    def backward_taped(self, grad_output: torch.Tensor, params=None):
        """``(grad_x, {name: gradient})`` for the tape of the last ``forward_train`` (``holo_unet_backward_taped``)."""
        tape = self.__dict__.pop("_holo_tape", None)
        if tape is None:
            raise _lib.HoloError("SimpleUnet3D.backward_taped: no taped forward (forward_train first)")
        x, t, h = tape
        dev = x.device
        L = runtime.lib()
        g = grad_output.contiguous().float()
        B = x.shape[0]
        if tuple(g.shape) != (B, self.out_channels) + tuple(x.shape[2:]):
            raise _lib.HoloError(f"grad_output must be {(B, self.out_channels) + tuple(x.shape[2:])}, got {tuple(g.shape)}")
        held = self.__dict__["_holo_train_ws"]
        gx = torch.empty_like(x)
        st = runtime.stream_ptr(dev)
        _lib.check(L, L.holo_unet_backward_taped(h, B, runtime.ptr(g), runtime.ptr(gx), runtime.ptr(held), held.numel(), st),
                   "holo_unet_backward_taped")
        grads = {}
        shapes = self.param_shapes()
        for k in (params if params is not None else self._param_names):
            out = torch.empty(shapes[k], device=dev)
            _lib.check(L, L.holo_unet_get_grad(h, k.encode(), runtime.ptr(out), out.numel(), runtime.ptr(held), st),
                       f"holo_unet_get_grad({k})")
            grads[k] = out
        return gx, grads

    def workspace_bytes(self, batch: int, device: torch.device) -> int:
        """Caller-owned HBM workspace of one forward at this batch size in the current compute mode (activations, GroupNorm
        partial sums, split-K / attention scratch)."""
        return int(runtime.lib().holo_unet_workspace_bytes(self._ensure_handle(device), batch))

    def fetch_block(self, tag: str, shape) -> torch.Tensor:
        """Debug/parity hook: output of block ``tag`` ("input_blocks.<i>", "middle_block", "output_blocks.<i>") of
        the LAST forward as an NCDHW tensor.  Needs HOLO_KEEP_INTERMEDIATES=1 when the net is created."""
        if self._handle is None:
            raise _lib.HoloError("fetch_block: no forward has run yet")
        dev = self._handle_device
        L = runtime.lib()
        ws = runtime.workspace(self, dev, 0)
        dst = torch.empty(tuple(shape), device=dev)
        n = C.c_int64()
        _lib.check(L, L.holo_unet_fetch_block(self._handle, tag.encode(), runtime.ptr(dst), dst.numel(), C.byref(n),
                                              runtime.ptr(ws), runtime.stream_ptr(dev)), f"fetch_block {tag}")
        if n.value != dst.numel():
            raise _lib.HoloError(f"fetch_block {tag}: {n.value} elements, expected {dst.numel()}")
        return dst

    # ---- measurement helper (bench.py) ----------------------------------------------------
    def time_convs(self, batch: int, iters: int, device: torch.device):
        """Average ms per forward spent in the conv3d implicit-GEMM launches, their FLOPs and launch count
        (hipEvents on the stream the kernels run on)."""
        h = self._ensure_handle(device)
        L = runtime.lib()
        nbytes = L.holo_unet_workspace_bytes(h, batch)
        ws = runtime.workspace(self, device, nbytes)
        ms, fl, nl = C.c_float(), C.c_double(), C.c_int()
        _lib.check(L, L.holo_unet_time_convs(h, batch, runtime.ptr(ws), ws.numel(), iters, runtime.stream_ptr(device),
                                             C.byref(ms), C.byref(fl), C.byref(nl)), "holo_unet_time_convs")
        return ms.value, fl.value, nl.value

    OP_NAMES = ("memset", "layout_in", "time_embed", "emb_linears", "gn_stats", "gn_finalize", "conv", "gemm",
                "softmax", "flash_attn", "layout_out")
    CONV_KERNELS = ("conv_igemm_kernel", "conv_halo_kernel", "conv_small_kernel", "conv_wino_kernel", "conv_wino2_kernel",
                    "conv_bf16t_kernel", "conv_wino3_kernel", "conv1x1_stream_kernel", "conv_bf16p_kernel", "conv_s2_bf16_kernel", "conv1x1_qkv_bf16_kernel", "conv1x1_bf16_stream_kernel")

    def time_ops(self, batch: int, iters: int, device: torch.device):
        """Per-op timing of one forward in execution order (hipEvents on the launch stream): list of dicts."""
        h = self._ensure_handle(device)
        L = runtime.lib()
        nbytes = L.holo_unet_workspace_bytes(h, batch)
        ws = runtime.workspace(self, device, nbytes)
        R = getattr(self, "_handle_size", None) or self.image_size
        x = torch.randn(batch, self.in_channels, R, R, R, device=device)
        y = torch.empty(batch, self.out_channels, R, R, R, device=device)
        t = torch.full((batch,), 500, dtype=torch.int64, device=device)
        cap = 1024
        arr = (_lib.HoloOpTiming * cap)()
        n = C.c_int()
        _lib.check(L, L.holo_unet_time_ops(h, batch, runtime.ptr(x), runtime.ptr(t), runtime.ptr(y), runtime.ptr(ws),
                                           ws.numel(), iters, runtime.stream_ptr(device), arr, cap, C.byref(n)),
                   "holo_unet_time_ops")
        ops = []
        for i in range(min(n.value, cap)):
            a = arr[i]
            d = dict(op=self.OP_NAMES[a.op], ms=a.ms, flops=a.flops, flops_executed=a.flops_executed or a.flops, cin=a.cin,
                     cout=a.cout, out_dim=a.out_dim)
            if a.op == 6:
                d.update(kernel=self.CONV_KERNELS[a.kernel], tile_depth=a.tile_depth, fused_skip=bool(a.fused_skip),
                         nsplit=a.nsplit, stride=a.stride, upsample=bool(a.upsample), ksz=a.ksz)
            ops.append(d)
        return ops


class _HoloUnetFn(torch.autograd.Function):
    """Autograd node of the HIP denoiser: forward = the taped forward (holo_unet_forward_train), backward =
    holo_unet_backward_taped when the tape is still this node's, else holo_unet_backward (which re-runs the forward with every
    intermediate kept).  The parameters are inputs of the node so that their ``.grad`` is filled."""

    @staticmethod
    def forward(ctx, net, x, timesteps, names, *params):
        ctx.net, ctx.names = net, names
        ctx.save_for_backward(x.detach(), timesteps.detach())
        ctx.x_needs = x.requires_grad
        ctx.tape_id = None
        with torch.no_grad():
            if net.compute_dtype == "f32" and getattr(net, "autograd_tape", True) and \
                    not os.environ.get("HOLO_NO_AUTOGRAD_TAPE"):
                # the taped forward: if nothing else uses the training workspace before this node's backward (another
                # differentiable call, an explicit backward()), that backward needs no second forward
                y = net.forward_train(x.detach(), timesteps)
                ctx.tape_id = net.__dict__["_holo_tape_id"]
                return y
            return net._forward_impl(x.detach(), timesteps)

    @staticmethod
    def backward(ctx, grad_y):
        x, t = ctx.saved_tensors
        want = [k for k, need in zip(ctx.names, ctx.needs_input_grad[4:]) if need]
        net = ctx.net
        if ctx.tape_id is not None and "_holo_tape" in net.__dict__ and net.__dict__.get("_holo_tape_id") == ctx.tape_id:
            gx, grads = net.backward_taped(grad_y, params=want)
        else:
            _, gx, grads = net.backward(x, t, grad_y, params=want)
        return (None, gx if ctx.x_needs else None, None, None) + tuple(grads.get(k) for k in ctx.names)
