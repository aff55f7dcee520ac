"""Render-side plugins backed by the fused HIP renderer.

Reference interfaces mirrored (paths relative to /root/reference/holo_diffusion):
  * ``RenderMLP`` (holo_voxel_grid_implicit_function.py:48-145): config fields :49-60, parameter names
    ``_density_net.mlp.<i>.0.{weight,bias}`` / ``_radiance_net.mlp.0.0.{weight,bias}``
    (custom_modules.py:94-113 wraps every Linear in a Sequential, hence the ``.0``)
  * ``HoloVoxelGridImplicitFunction`` (:148-269): config fields :150-160, ``allows_multiple_passes``,
    nested ``render_mlp`` built with ``input_dims = n_hidden`` (:165-172)
  * ``ImplicitFunctionWrapper.bind_args / unbind_args`` as used at holo_diffusion_model.py:166-168,437-438,466-467
  * ``HoloMultiPassEmissionAbsorptionRenderer`` (holo_multipass_ea.py:15-125) with the config fields of
    configs/apple.yaml:147-165; ``forward(ray_bundle, implicit_functions, evaluation_mode) -> RendererOutput``
  * ``AdaptiveRaySampler`` call at holo_diffusion_model.py:442-448 (configs/apple.yaml:135-146)

All arithmetic (ray generation, voxel fetch, RenderMLP, emission-absorption compositing, importance
resampling, fine pass) happens inside ``holo_render`` (csrc/kernels_render.hip); the classes here carry
configuration and parameters and keep the call structure of the reference.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _lib, runtime
from .cameras import PerspectiveCameras
from .registry import Configurable, apply_config, pt3d_base, registry

COLOUR_DIMS = 3


class EvaluationMode(enum.Enum):
    TRAINING = "training"
    EVALUATION = "evaluation"


class RenderSamplingMode(enum.Enum):
    MASK_SAMPLE = "mask_sample"
    FULL_GRID = "full_grid"


class HiddenActivation(enum.Enum):
    RELU = "relu"
    SOFTPLUS = "softplus"
    LEAKYRELU = "leakyrelu"


@dataclass
class ImplicitronRayBundle:
    """Full-grid ray bundle.  The fused kernel regenerates origins/directions/lengths from the camera, so the
    tensors are only materialised on request (``materialize()``), e.g. for inspection."""
    camera: PerspectiveCameras
    image_height: int
    image_width: int
    n_pts_per_ray: int
    scene_extent: float
    scene_center: Tuple[float, float, float]
    origins: Optional[torch.Tensor] = None
    directions: Optional[torch.Tensor] = None
    lengths: Optional[torch.Tensor] = None
    xys: Optional[torch.Tensor] = None
    # training mode: `xys` (n_cam, n_rays, 1, 2) is the explicit ray list (mask-sampled rays); `stratified` asks the
    # renderer for stratified depths (stratified_point_sampling_training)
    training: bool = False
    stratified: bool = False

    def materialize(self) -> "ImplicitronRayBundle":
        """Fill origins (n,H,W,3), directions (n,H,W,3), lengths (n,H,W,P), xys (n,H,W,2) with plain torch ops
        (PyTorch3D NDCMultinomialRaysampler/_xy_to_ray_bundle + AdaptiveRaySampler bounds).  The fused renderer
        never needs these tensors; they exist for inspection and for the stand-alone implicit function."""
        cams, H, W = self.camera, self.image_height, self.image_width
        dev = cams.R.device
        rx, ry = (W / H, 1.0) if W >= H else (1.0, H / W)
        xs = torch.linspace(rx - rx / W, -rx + rx / W, W, dtype=torch.float32, device=dev)
        ys = torch.linspace(ry - ry / H, -ry + ry / H, H, dtype=torch.float32, device=dev)
        Y, X = torch.meshgrid(ys, xs, indexing="ij")
        xy = torch.stack([X, Y], dim=-1).reshape(1, -1, 2)
        f, pp = cams.focal_xy()[:, None, :], cams.principal_point[:, None, :]
        d_cam = torch.cat([(xy - pp) / f, torch.ones_like(xy[..., :1]).expand(len(cams), -1, -1)], dim=-1)
        Rt = cams.R.transpose(1, 2)
        p1 = torch.bmm(d_cam - cams.T[:, None, :], Rt)
        p2 = torch.bmm(2.0 * d_cam - cams.T[:, None, :], Rt)
        dirs = p2 - p1
        centre = cams.get_camera_center()
        sc = torch.tensor(self.scene_center, dtype=torch.float32, device=dev)
        dist = ((centre - sc) ** 2).sum(-1).clamp(0.001).sqrt().clamp(self.scene_extent + 1e-3)
        n = len(cams)
        self.origins = (p1 - dirs).reshape(n, H, W, 3)
        self.directions = dirs.reshape(n, H, W, 3)
        self.lengths = torch.stack([torch.linspace(float(d - self.scene_extent), float(d + self.scene_extent),
                                                   self.n_pts_per_ray, dtype=torch.float32, device=dev)
                                    for d in dist])[:, None, None, :].expand(n, H, W, -1).contiguous()
        self.xys = xy.reshape(1, H, W, 2).expand(n, -1, -1, -1)
        return self


@dataclass
class RendererOutput:
    features: torch.Tensor
    depths: torch.Tensor
    masks: torch.Tensor
    prev_stage: Optional["RendererOutput"] = None
    normals: Optional[torch.Tensor] = None
    weights: Optional[torch.Tensor] = None
    aux: Dict[str, Any] = field(default_factory=dict)


class _Node(torch.nn.Module):
    pass


def _add_param(root: torch.nn.Module, dotted: str, p: torch.nn.Parameter) -> None:
    parts = dotted.split(".")
    m = root
    for part in parts[:-1]:
        if part not in m._modules:
            m.add_module(part, _Node())
        m = m._modules[part]
    m.register_parameter(parts[-1], p)


class RenderMLP(Configurable, torch.nn.Module):
    input_dims: int = 128
    output_feature_dims: int = COLOUR_DIMS
    output_vp_independent_feature_dims: int = 64
    feat_emb_dims: int = 0
    dir_emb_dims: int = 4
    dnet_num_layers: int = 4
    dnet_hidden_dim: int = 256
    dnet_input_skips: Tuple[int, ...] = (2,)
    rnet_num_layers: int = 1
    rnet_hidden_dim: int = 128
    rnet_input_skips: Tuple[int, ...] = ()
    activation_fn: HiddenActivation = HiddenActivation.LEAKYRELU

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        apply_config(self, kwargs)
        if isinstance(self.activation_fn, str):
            self.activation_fn = HiddenActivation[self.activation_fn]
        unsupported = []
        if self.feat_emb_dims != 0:
            unsupported.append("feat_emb_dims != 0")
        if self.dnet_num_layers != 4 or tuple(self.dnet_input_skips) != (2,):
            unsupported.append("density net other than 4 layers with input skip at layer 2")
        if self.rnet_num_layers != 1 or tuple(self.rnet_input_skips) != ():
            unsupported.append("radiance net other than a single layer")
        if self.activation_fn != HiddenActivation.LEAKYRELU:
            unsupported.append("activation_fn other than LEAKYRELU")
        if self.output_feature_dims != COLOUR_DIMS:
            unsupported.append("output_feature_dims other than the 3 colour channels")
        if self.output_vp_independent_feature_dims < 0 or self.output_vp_independent_feature_dims % 4:
            unsupported.append("output_vp_independent_feature_dims that is not a multiple of 4")
        if self.dnet_hidden_dim != 256 or self.dir_emb_dims != 4:
            unsupported.append("dnet_hidden_dim other than 256 / dir_emb_dims other than 4")
        if self.input_dims not in (16, 32, 64, 128):
            unsupported.append("input_dims other than 16, 32, 64 or 128")
        if unsupported:
            raise NotImplementedError("RenderMLP: the fused renderer supports the released configuration only; got "
                                      + "; ".join(unsupported))
        self._density_net = _Node()
        self._radiance_net = _Node()
        self._feature_net = _Node() if self.output_vp_independent_feature_dims > 0 else None
        for name, shape in self.param_shapes().items():
            w = torch.empty(shape)
            if name.endswith("weight"):
                torch.nn.init.xavier_uniform_(w)  # _xavier_init (custom_modules.py:104-105)
            else:
                w.zero_()
            _add_param(self, name, torch.nn.Parameter(w, requires_grad=False))

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        Cf, Hd = self.input_dims, self.dnet_hidden_dim
        demb = 3 * (2 * self.dir_emb_dims + 1)
        shapes = {
            "_density_net.mlp.0.0.weight": (Hd, Cf), "_density_net.mlp.0.0.bias": (Hd,),
            "_density_net.mlp.1.0.weight": (Hd, Hd), "_density_net.mlp.1.0.bias": (Hd,),
            "_density_net.mlp.2.0.weight": (Hd, Hd + Cf), "_density_net.mlp.2.0.bias": (Hd,),
            "_density_net.mlp.3.0.weight": (Hd + 1, Hd), "_density_net.mlp.3.0.bias": (Hd + 1,),
            "_radiance_net.mlp.0.0.weight": (COLOUR_DIMS, Hd + demb), "_radiance_net.mlp.0.0.bias": (COLOUR_DIMS,),
        }
        if self.output_vp_independent_feature_dims > 0:  # the feature head (holo_voxel_grid_implicit_function.py:94-105)
            Fd = self.output_vp_independent_feature_dims
            shapes["_feature_net.mlp.0.0.weight"] = (Fd, Hd)
            shapes["_feature_net.mlp.0.0.bias"] = (Fd,)
        return shapes

    def forward(self, features: torch.Tensor, view_dirs: torch.Tensor):
        """``RenderMLP.forward(features, view_dirs) -> (densities, radiance, vp_independent_features | None)``
        (holo_voxel_grid_implicit_function.py:107-129) for features that are ALREADY sampled, (..., input_dims), with one
        unit view direction per feature row, (..., 3) - the call of the reference's tests/test_voxel_grid_implicit_function.py
        ::test_RenderMLP_forward.  Runs the same kernel as the implicit function on a one-voxel-per-row grid stand-in:
        the features are handed over as an (n, C) "grid" whose trilinear fetch at the voxel centres is the identity."""
        return _render_mlp_forward(self, features, view_dirs)


class _NativeRenderMlp:
    """Owns a HoloRenderer handle for one (implicit function, render config) and keeps the packed RenderMLP in
    sync with the torch parameters (re-committed when a parameter's storage or version changes)."""

    def __init__(self):
        self.handle: Optional[C.c_void_p] = None
        self.key = None
        self.versions = None

    def ensure(self, device, mlp: "RenderMLP", cfg_kwargs: dict) -> C.c_void_p:
        L = runtime.lib()
        key = (device,) + tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v)
                                       for k, v in cfg_kwargs.items()))
        params = dict(mlp.named_parameters())
        versions = tuple((k, p.data_ptr(), p._version) for k, p in params.items())
        if self.handle is None or key != self.key:
            self.close()
            cfg = _lib.make_render_cfg(**cfg_kwargs)
            h = C.c_void_p()
            _lib.check(L, L.holo_renderer_create(runtime.ctx(device), C.byref(cfg), C.byref(h)), "holo_renderer_create")
            self.handle, self.key, self.versions = h, key, None
        if versions != self.versions:
            st = runtime.stream_ptr(device)
            for k, p in params.items():
                if p.device != device or p.dtype != torch.float32:
                    raise _lib.HoloError(f"RenderMLP parameter '{k}' is {p.dtype} on {p.device}; expected float32 on {device}")
                t = p.detach().contiguous()
                _lib.check(L, L.holo_renderer_set_param(self.handle, k.encode(), runtime.ptr(t), _lib.HOLO_DTYPE_F32,
                                                       t.dim(), _lib.shape_array(t.shape), st), f"set_param({k})")
            _lib.check(L, L.holo_renderer_commit(self.handle, st), "holo_renderer_commit")
            self.versions = versions
        return self.handle

    def close(self):
        if self.handle is not None:
            try:
                runtime.lib().holo_renderer_destroy(self.handle)
            except Exception:
                pass
            self.handle = None

    def __del__(self):
        self.close()


def _render_mlp_forward(mlp: "RenderMLP", features: torch.Tensor, view_dirs: torch.Tensor):
    """RenderMLP on already-sampled features through the implicit-function kernel.  The rows are laid out as the voxels
    of an R^3 grid with R - 1 a power of two, volume_extent = R (voxel size 1, half extent 2^k) and evaluated at the voxel
    centres i - (R-1)/2: every coordinate, the local coordinate and the voxel index are then EXACT in fp32, so the
    trilinear fetch returns the row itself (weights exactly 1 and 0)."""
    runtime.require_device(features, "RenderMLP.forward")
    dev = features.device
    Cf, Fd = mlp.input_dims, int(mlp.output_vp_independent_feature_dims)
    lead = tuple(features.shape[:-1])
    f = features.reshape(-1, Cf).float()
    d = view_dirs.expand(*lead, 3).reshape(-1, 3).float().contiguous()
    n = f.shape[0]
    R = next(r for r in (3, 5, 9, 17, 33) if r ** 3 >= n or r == 33)
    if "_native_rows" not in mlp.__dict__:
        mlp.__dict__["_native_rows"] = _NativeRenderMlp()
    h = mlp.__dict__["_native_rows"].ensure(dev, mlp, dict(resol=R, feature_size=Cf, image_height=8, image_width=8,
                                                           volume_extent=float(R), dnet_hidden_dim=mlp.dnet_hidden_dim,
                                                           dir_emb_dims=mlp.dir_emb_dims, feature_dim=Fd))
    L = runtime.lib()
    dens, col = torch.empty(n, device=dev), torch.empty(n, 3, device=dev)
    vp = torch.empty(n, Fd, device=dev) if Fd > 0 else None
    cap = R ** 3
    idx = torch.arange(min(n, cap), device=dev)
    pts_all = torch.stack([idx % R, (idx // R) % R, idx // (R * R)], dim=-1).float() - 0.5 * (R - 1)
    grid = torch.zeros(1, Cf, R, R, R, device=dev)
    for s0 in range(0, n, cap):
        m = min(cap, n - s0)
        grid.view(Cf, -1)[:, :m] = f[s0:s0 + m].t()
        pts = pts_all[:m].contiguous()
        dd = d[s0:s0 + m].contiguous()
        ws = runtime.workspace(mlp, dev, L.holo_implicit_workspace_bytes(h, m, 1, 1 if Fd > 0 else 0))
        _lib.check(L, L.holo_implicit_eval_features(
            h, runtime.ptr(grid), runtime.ptr(pts), runtime.ptr(dd), m, 1, runtime.ptr(dens[s0:]), runtime.ptr(col[s0:]),
            runtime.ptr(vp[s0:]) if vp is not None else C.c_void_p(None), runtime.ptr(ws), ws.numel(),
            runtime.stream_ptr(dev)), "holo_implicit_eval_features")
    return dens.reshape(*lead, 1), col.reshape(*lead, 3), (vp.reshape(*lead, Fd) if vp is not None else None)


def _camera_array(cams):
    """HoloCamera[n] launch parameters from a camera batch, built on the host.  ``PerspectiveCameras`` of this package
    carry a cached host copy (no device synchronisation); any other object with PyTorch3D's ``R, T, focal_length,
    principal_point`` attributes is accepted and read back with one ``.cpu()`` each."""
    if hasattr(cams, "host"):
        Rc, Tc, fc, pc = cams.host()
    else:
        def cpu(x, last):
            return torch.as_tensor(x).detach().to("cpu", torch.float32).reshape(-1, last)
        Rc = cpu(cams.R, 9)
        n = Rc.shape[0]
        f = torch.as_tensor(cams.focal_length)
        Tc, pc = cpu(cams.T, 3).expand(n, 3), cpu(cams.principal_point, 2).expand(n, 2)
        fc = cpu(f, 2 if (f.dim() >= 1 and f.shape[-1] == 2) else 1).expand(n, -1)
        fc = fc.expand(n, 2) if fc.shape[1] == 1 else fc
    n_cam = Rc.shape[0]
    arr = (_lib.HoloCamera * n_cam)()
    Rl, Tl, fl, pl = Rc.reshape(n_cam, 9).tolist(), Tc.tolist(), fc.tolist(), pc.tolist()
    for i in range(n_cam):
        arr[i].R[:] = Rl[i]
        arr[i].T[:] = Tl[i]
        arr[i].focal[:] = fl[i]
        arr[i].principal_point[:] = pl[i]
    return arr


# PyTorch3D's own plugin bases when importable (registry keys of Implicitron's factories), stand-ins otherwise
ImplicitFunctionBase = pt3d_base("implicitron.models.implicit_function.base", "ImplicitFunctionBase")


@registry.register
class HoloVoxelGridImplicitFunction(ImplicitFunctionBase, torch.nn.Module):
    resol: int = 32
    volume_extent: float = 8.0
    n_hidden: int = 128
    feature_dim: int = 64
    init_density_bias: float = 1e-4
    render_normals: bool = False
    render_mlp_args: Optional[dict] = None

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        apply_config(self, kwargs)
        self.create_render_mlp()

    def create_render_mlp(self):
        args = dict(self.render_mlp_args or {})
        args.update({"input_dims": self.n_hidden, "output_feature_dims": COLOUR_DIMS,
                     "output_vp_independent_feature_dims": self.feature_dim})
        self.render_mlp = RenderMLP(**args)

    @staticmethod
    def allows_multiple_passes() -> bool:
        return True

    def forward(self, *, ray_bundle=None, fun_viewpool=None, camera=None, global_code=None, run_id=None,
                pass_number=None, pts_3d=None, voxel_grid_features=None, **kwargs):
        """Stand-alone evaluation (holo_voxel_grid_implicit_function.py:182-269): trilinear fetch of the voxel
        grid at the ray points (or at ``pts_3d``) + RenderMLP -> ``(densities (...,1), features (...,3), aux)``.
        The fused renderer does not go through this method; it exists for drop-in parity with the reference
        (and its tests, which call it with ``pts_3d``)."""
        assert voxel_grid_features is not None, "voxel_grid_features must be provided!"
        assert ray_bundle is not None or pts_3d is not None, "either ray_bundle or pts_3d must be provided!"
        # render_normals (released YAMLs set it, configs/apple.yaml:203): the stand-alone evaluation returns
        # aux["normals"] (below); inside the fused renderer the normals are not rendered - HoloDiffusionModel drops them
        # (no `normals_render` output).
        if pts_3d is None:
            if ray_bundle.origins is None:
                ray_bundle.materialize()
            pts = (ray_bundle.origins[..., None, :]
                   + ray_bundle.lengths[..., :, None] * ray_bundle.directions[..., None, :])
            dirs = ray_bundle.directions
        else:
            pts = pts_3d
            dirs = torch.ones(*pts.shape[:-2], 3, dtype=torch.float32, device=pts.device)  # dummy directions (:229-236)
        runtime.require_device(pts, "HoloVoxelGridImplicitFunction.forward")
        dev = pts.device
        grid = voxel_grid_features
        if tuple(grid.shape) != (1, self.n_hidden, self.resol, self.resol, self.resol):
            raise _lib.HoloError(f"voxel grid must be (1,{self.n_hidden},{self.resol}^3), got {tuple(grid.shape)}")
        spatial = tuple(pts.shape[:-1])
        per_dir = spatial[-1]
        ptsf = pts.reshape(-1, 3).contiguous().float()
        dirsf = dirs.reshape(-1, 3).contiguous().float()
        n = ptsf.shape[0]
        if dirsf.shape[0] * per_dir != n:
            raise _lib.HoloError("directions do not match the ray points")
        if not hasattr(self, "_native"):
            self._native = _NativeRenderMlp()
        h = self._native.ensure(dev, self.render_mlp, dict(
            resol=self.resol, feature_size=self.n_hidden, image_height=8, image_width=8,
            volume_extent=float(self.volume_extent), dnet_hidden_dim=self.render_mlp.dnet_hidden_dim,
            dir_emb_dims=self.render_mlp.dir_emb_dims, feature_dim=self.feature_dim))
        L = runtime.lib()
        dens = torch.empty(n, device=dev)
        col = torch.empty(n, 3, device=dev)
        Fd = int(self.feature_dim)
        vp = torch.empty(n, Fd, device=dev) if Fd > 0 else None
        nbytes = max(L.holo_implicit_workspace_bytes(h, n, per_dir, 1 if Fd > 0 else 0), L.holo_render_workspace_bytes(h, 1, 0))
        ws = runtime.workspace(self, dev, nbytes)
        _lib.check(L, L.holo_implicit_eval_features(
            h, runtime.ptr(grid.contiguous().float()), runtime.ptr(ptsf), runtime.ptr(dirsf), n, per_dir, runtime.ptr(dens),
            runtime.ptr(col), runtime.ptr(vp) if vp is not None else C.c_void_p(None), runtime.ptr(ws), ws.numel(),
            runtime.stream_ptr(dev)), "holo_implicit_eval_features")
        aux = {}
        if self.render_normals:  # RenderMLP.get_normals (:131-145), evaluated analytically in the kernel
            nrm = torch.empty(n, 3, device=dev)
            _lib.check(L, L.holo_implicit_normals(h, runtime.ptr(grid.contiguous().float()), runtime.ptr(ptsf), n,
                                                  runtime.ptr(nrm), runtime.ptr(ws), ws.numel(),
                                                  runtime.stream_ptr(dev)), "holo_implicit_normals")
            aux["normals"] = nrm.reshape(*spatial, 3)
        features = col.reshape(*spatial, COLOUR_DIMS)
        if vp is not None:  # features = cat(colour, view-point independent features) (:265-269)
            features = torch.cat([features, vp.reshape(*spatial, Fd)], dim=-1)
        return dens.reshape(*spatial, 1), features, aux


class ImplicitFunctionWrapper(torch.nn.Module):
    def __init__(self, fn: torch.nn.Module):
        super().__init__()
        self._fn = fn
        self.bound_args: Dict[str, Any] = {}

    def bind_args(self, **bound_args):
        self.bound_args = bound_args
        self._fn.on_bind_args() if hasattr(self._fn, "on_bind_args") else None

    def unbind_args(self):
        self.bound_args = {}

    def forward(self, *args, **kwargs):
        return self._fn(*args, **{**kwargs, **self.bound_args})


class AdaptiveRaySampler(Configurable):
    """Evaluation-mode (full grid, no stratification) subset of PyTorch3D's AdaptiveRaySampler."""
    image_width: int = 400
    image_height: int = 400
    n_pts_per_ray_training: int = 64
    n_pts_per_ray_evaluation: int = 64
    n_rays_per_image_sampled_from_mask: int = 1024
    n_rays_total_training: Optional[int] = None
    stratified_point_sampling_training: bool = True
    stratified_point_sampling_evaluation: bool = False
    scene_extent: float = 8.0
    scene_center: Tuple[float, float, float] = (0.0, 0.0, 0.0)

    def __init__(self, **kwargs):
        apply_config(self, kwargs)

    def __call__(self, cameras: PerspectiveCameras, evaluation_mode: EvaluationMode, mask=None,
                 sampling_mode: Optional[RenderSamplingMode] = None, xys: Optional[torch.Tensor] = None) -> ImplicitronRayBundle:
        """EVALUATION: the full pixel grid (regenerated inside the kernel).  TRAINING: ``mask_sample`` draws
        ``n_rays_per_image_sampled_from_mask`` pixels per camera from the (nearest-resized) mask as a multinomial
        distribution, without replacement while the mask has enough support (PyTorch3D MultinomialRaysampler._sample_mask /
        _safe_multinomial, restated - a random draw, UNPINNED by nature); ``xys`` (n_cam, n_rays, 2) overrides the draw."""
        training = evaluation_mode == EvaluationMode.TRAINING
        if not training:
            if self.stratified_point_sampling_evaluation:
                raise NotImplementedError("stratified sampling at evaluation time is not supported")
            return ImplicitronRayBundle(camera=cameras, image_height=self.image_height, image_width=self.image_width,
                                        n_pts_per_ray=self.n_pts_per_ray_evaluation, scene_extent=self.scene_extent,
                                        scene_center=tuple(self.scene_center))
        n_cam = len(cameras)
        H, W = self.image_height, self.image_width
        if xys is None:
            if sampling_mode == RenderSamplingMode.MASK_SAMPLE and mask is not None:
                n_rays = self.n_rays_per_image_sampled_from_mask
                wts = torch.nn.functional.interpolate(mask.float(), size=[H, W], mode="nearest").reshape(n_cam, -1)
                enough = (wts > 0).sum(dim=1) >= n_rays
                idx = torch.stack([torch.multinomial(wts[i] if bool(wts[i].sum() > 0) else torch.ones_like(wts[i]), n_rays,
                                                     replacement=not bool(enough[i])) for i in range(n_cam)])
            else:  # full grid in training mode
                idx = torch.arange(H * W, device=cameras.R.device)[None].expand(n_cam, -1)
            rx, ry = (W / H, 1.0) if W >= H else (1.0, H / W)
            xs = torch.linspace(rx - rx / W, -rx + rx / W, W, dtype=torch.float32, device=idx.device)
            ys = torch.linspace(ry - ry / H, -ry + ry / H, H, dtype=torch.float32, device=idx.device)
            xys = torch.stack([xs[idx % W], ys[idx // W]], dim=-1)
        xys = xys.reshape(n_cam, -1, 1, 2).float()
        return ImplicitronRayBundle(camera=cameras, image_height=H, image_width=W, n_pts_per_ray=self.n_pts_per_ray_training,
                                    scene_extent=self.scene_extent, scene_center=tuple(self.scene_center), xys=xys,
                                    training=True, stratified=bool(self.stratified_point_sampling_training))


BaseRenderer = pt3d_base("implicitron.models.renderer.base", "BaseRenderer")


@registry.register
class HoloMultiPassEmissionAbsorptionRenderer(BaseRenderer, torch.nn.Module):
    raymarcher_class_type: str = "EmissionAbsorptionRaymarcher"
    n_pts_per_ray_fine_training: int = 64
    n_pts_per_ray_fine_evaluation: int = 64
    stratified_sampling_coarse_training: bool = True
    stratified_sampling_coarse_evaluation: bool = False
    append_coarse_samples_to_fine: bool = True
    density_noise_std_train: float = 1.0
    return_weights: bool = False
    raymarcher_EmissionAbsorptionRaymarcher_args: Optional[dict] = None
    # build-side extension (not a reference field): "f32" (exact-fp32 MFMA) or "f32_bf16x3" (RenderMLP products from an
    # exact three-term bf16 split on the bf16 matrix cores, fp32 accumulation) - holo_renderer_set_compute_dtype
    compute_dtype: str = "f32"

    _RAYMARCHER_DEFAULTS = dict(surface_thickness=1, bg_color=(0.0,), replicate_last_interval=False,
                                background_opacity=1e10, density_relu=True, blend_output=False)

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        apply_config(self, kwargs)
        rm = dict(self._RAYMARCHER_DEFAULTS)
        rm.update(self.raymarcher_EmissionAbsorptionRaymarcher_args or {})
        self._raymarcher_args = rm
        bad = []
        if self.raymarcher_class_type != "EmissionAbsorptionRaymarcher":
            bad.append("raymarcher_class_type")
        if rm["surface_thickness"] != 1 or rm["replicate_last_interval"] or not rm["density_relu"] or rm["blend_output"]:
            bad.append("raymarcher args other than the released ones (configs/apple.yaml:156-165)")
        if not self.append_coarse_samples_to_fine or self.stratified_sampling_coarse_evaluation or self.return_weights:
            bad.append("refiner/weights options other than the released ones (configs/apple.yaml:147-155)")
        if bad:
            raise NotImplementedError("HoloMultiPassEmissionAbsorptionRenderer: unsupported " + "; ".join(bad))
        self._handle: Optional[C.c_void_p] = None
        self._handle_key = None
        self._param_versions = None

    def _bg_color(self) -> Tuple[float, float, float]:
        bg = tuple(float(v) for v in self._raymarcher_args["bg_color"])
        return bg * 3 if len(bg) == 1 else bg

    def _ensure_handle(self, fn: HoloVoxelGridImplicitFunction, bundle: ImplicitronRayBundle, device) -> C.c_void_p:
        L = runtime.lib()
        mlp = fn.render_mlp
        key = (device, fn.resol, fn.n_hidden, float(fn.volume_extent), bundle.image_height, bundle.image_width,
               bundle.n_pts_per_ray, self.n_pts_per_ray_fine_evaluation, float(bundle.scene_extent),
               tuple(bundle.scene_center), self._bg_color(), float(self._raymarcher_args["background_opacity"]),
               mlp.dnet_hidden_dim, mlp.dir_emb_dims)
        params = dict(mlp.named_parameters())
        versions = tuple((k, p.data_ptr(), p._version) for k, p in params.items())
        if self._handle is None or key != self._handle_key:
            if self._handle is not None:
                L.holo_renderer_destroy(self._handle)
            cfg = _lib.make_render_cfg(fn.resol, fn.n_hidden, bundle.image_height, bundle.image_width,
                                       volume_extent=fn.volume_extent, scene_extent=bundle.scene_extent,
                                       scene_center=bundle.scene_center, n_pts_coarse=bundle.n_pts_per_ray,
                                       n_pts_fine=self.n_pts_per_ray_fine_evaluation, bg_color=self._bg_color(),
                                       background_opacity=self._raymarcher_args["background_opacity"],
                                       dnet_hidden_dim=mlp.dnet_hidden_dim, dir_emb_dims=mlp.dir_emb_dims)
            h = C.c_void_p()
            _lib.check(L, L.holo_renderer_create(runtime.ctx(device), C.byref(cfg), C.byref(h)), "holo_renderer_create")
            self._handle, self._handle_key, self._param_versions = h, key, None
        if versions != self._param_versions:
            st = runtime.stream_ptr(device)
            for k, p in params.items():
                if p.device != device or p.dtype != torch.float32:
                    raise _lib.HoloError(f"RenderMLP parameter '{k}' is {p.dtype} on {p.device}; expected float32 on {device}")
                t = p.detach().contiguous()
                _lib.check(L, L.holo_renderer_set_param(self._handle, k.encode(), runtime.ptr(t), _lib.HOLO_DTYPE_F32,
                                                       t.dim(), _lib.shape_array(t.shape), st), f"set_param({k})")
            _lib.check(L, L.holo_renderer_commit(self._handle, st), "holo_renderer_commit")
            self._param_versions = versions
        code = {"f32": _lib.HOLO_DTYPE_F32, "f32_bf16x3": _lib.HOLO_DTYPE_F32_BF16X3}.get(self.compute_dtype)
        if code is None:
            raise _lib.HoloError(f"renderer compute_dtype must be 'f32' or 'f32_bf16x3' (got {self.compute_dtype!r})")
        _lib.check(L, L.holo_renderer_set_compute_dtype(self._handle, code), "holo_renderer_set_compute_dtype")
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                runtime.lib().holo_renderer_destroy(self._handle)
        except Exception:
            pass

    def _training_setup(self, bundle: ImplicitronRayBundle, implicit_functions, rng_streams, draw: bool = True):
        """Arguments shared by the training-mode forward and its backward: handle (keyed on the training sample counts),
        grid, ray list, random streams (``draw = False``: every stream in use must be injected - the backward pass has to
        see the draws of the forward pass)."""
        if not bundle.training or bundle.xys is None:
            raise ValueError("training-mode rendering needs the ray sampler's training bundle (explicit xys)")
        wrapper = implicit_functions[0]
        fn = wrapper._fn
        grid = wrapper.bound_args.get("voxel_grid_features")
        if grid is None:
            raise ValueError("voxel_grid_features must be bound to the implicit function (bind_args)")
        runtime.require_device(grid, "HoloMultiPassEmissionAbsorptionRenderer.forward")
        dev = grid.device
        if not isinstance(fn, HoloVoxelGridImplicitFunction) or fn.feature_dim != 0 or fn.n_hidden not in (16, 32, 64):
            raise NotImplementedError("training-mode rendering: HoloVoxelGridImplicitFunction with 16/32/64 grid features, colours only")
        cams = bundle.camera
        n_cam, n_rays = len(cams), int(bundle.xys.shape[1])
        P, Pf = int(bundle.n_pts_per_ray), int(self.n_pts_per_ray_fine_training)
        two_pass = len(implicit_functions) > 1
        # the handle is keyed on the sample counts: a training handle next to the evaluation one
        eval_bundle = ImplicitronRayBundle(camera=cams, image_height=bundle.image_height, image_width=bundle.image_width,
                                           n_pts_per_ray=P, scene_extent=bundle.scene_extent, scene_center=bundle.scene_center)
        saved = (self._handle, self._handle_key, self._param_versions, self.n_pts_per_ray_fine_evaluation)
        if "_train_state" not in self.__dict__:
            self.__dict__["_train_state"] = (None, None, None)
        self._handle, self._handle_key, self._param_versions = self.__dict__["_train_state"]
        self.n_pts_per_ray_fine_evaluation = Pf
        try:
            h = self._ensure_handle(fn, eval_bundle, dev)
        finally:
            self.__dict__["_train_state"] = (self._handle, self._handle_key, self._param_versions)
            self._handle, self._handle_key, self._param_versions, self.n_pts_per_ray_fine_evaluation = saved
        rs = dict(rng_streams or {})
        std = float(self.density_noise_std_train)

        def stream(key, shape, normal, wanted):
            if not wanted:
                return None
            t = rs.get(key)
            if t is None:
                if not draw:
                    raise _lib.HoloError(f"rng_streams['{key}'] is needed: the backward pass takes the draws of the forward pass")
                t = torch.randn(shape, device=dev) if normal else torch.rand(shape, device=dev)
            if tuple(t.shape) != tuple(shape):
                raise _lib.HoloError(f"rng_streams['{key}'] must have shape {tuple(shape)}, got {tuple(t.shape)}")
            return t.to(dev, torch.float32).contiguous()

        # the ray sampler's flag alone jitters the coarse depths; the renderer's flag is the RayPointRefiner's
        # `random_sampling` (fine pass) only (PyTorch3D MultiPassEmissionAbsorptionRenderer.__post_init__)
        u_c = stream("u_coarse", (n_cam, n_rays, P), False, bundle.stratified)
        u_f = stream("u_fine", (n_cam, n_rays, Pf), False, two_pass and self.stratified_sampling_coarse_training)
        nz_c = stream("noise_coarse", (n_cam, n_rays, P), True, std > 0.0)
        nz_f = stream("noise_fine", (n_cam, n_rays, P + Pf), True, std > 0.0 and two_pass)
        xys = bundle.xys.reshape(n_cam, n_rays, 2).to(dev, torch.float32).contiguous()
        grid = grid.contiguous().float()
        return dict(h=h, fn=fn, grid=grid, dev=dev, cams=cams, n_cam=n_cam, n_rays=n_rays, P=P, Pf=Pf, two_pass=two_pass,
                    u_c=u_c, u_f=u_f, nz_c=nz_c, nz_f=nz_f, std=std, xys=xys)

    def _forward_training(self, bundle: ImplicitronRayBundle, implicit_functions, rng_streams) -> RendererOutput:
        """Training-mode forward of the two-pass renderer (holo_multipass_ea.py:79-125 with
        evaluation_mode = TRAINING): explicit ray list, ``n_pts_per_ray_training`` coarse + ``n_pts_per_ray_fine_training``
        new samples, stratified depths / importance samples, density noise of std ``density_noise_std_train`` on both
        passes.  The random streams are drawn with torch on the device unless injected through ``rng_streams`` (keys
        ``u_coarse (n_cam,n_rays,P)``, ``u_fine (n_cam,n_rays,Pf)``, ``noise_coarse (n_cam,n_rays,P)``,
        ``noise_fine (n_cam,n_rays,P+Pf)`` - the parity tests inject them).  With autograd enabled and a grid or RenderMLP
        parameter that requires grad the outputs hang on an autograd node (``_HoloRenderRaysFn``) whose backward is
        ``holo_render_rays_backward``: ``loss.backward()`` fills the grid's and the parameters' ``.grad``."""
        a = self._training_setup(bundle, implicit_functions, rng_streams)
        mlp_params = dict(a["fn"].render_mlp.named_parameters())
        grid_in = implicit_functions[0].bound_args.get("voxel_grid_features")
        needs_grad = torch.is_grad_enabled() and (grid_in.requires_grad or any(p.requires_grad for p in mlp_params.values()))
        if needs_grad and not a["two_pass"]:
            raise NotImplementedError("the differentiable training-mode renderer is the two-pass one (coarse + fine implicit "
                                      "function, holo_render_rays_backward); a single-pass render would return outputs "
                                      "detached from the graph")
        if needs_grad:
            # the draws of this call (injected or made by _training_setup) are what the backward pass must see again
            streams = {k: a[s] for k, s in (("u_coarse", "u_c"), ("u_fine", "u_f"), ("noise_coarse", "nz_c"), ("noise_fine", "nz_f"))
                       if a[s] is not None}
            names = list(mlp_params)
            outs = _HoloRenderRaysFn.apply(self, bundle, implicit_functions, streams, names, grid_in, *[mlp_params[k] for k in names])
            n_cam, n_rays = a["n_cam"], a["n_rays"]
            shp = lambda t, c: t.reshape(n_cam, c, n_rays, 1).permute(0, 2, 3, 1)  # noqa: E731
            coarse = RendererOutput(features=shp(outs[3], 3), depths=shp(outs[4], 1), masks=shp(outs[5], 1))
            return RendererOutput(features=shp(outs[0], 3), depths=shp(outs[1], 1), masks=shp(outs[2], 1), prev_stage=coarse)
        return self._render_rays_raw(a, as_output=True)

    def _render_rays_raw(self, a: dict, as_output: bool):
        """holo_render_rays on the arguments of ``_training_setup``: the six raw planes (img (n_cam,3,n_rays), dep, msk and the
        coarse three) or the RendererOutput built from them."""
        h, grid, dev, cams, n_cam, n_rays, two_pass = a["h"], a["grid"], a["dev"], a["cams"], a["n_cam"], a["n_rays"], a["two_pass"]
        u_c, u_f, nz_c, nz_f, std, xys = a["u_c"], a["u_f"], a["nz_c"], a["nz_f"], a["std"], a["xys"]
        L = runtime.lib()
        img = torch.empty(n_cam, 3, n_rays, device=dev)
        dep, msk = torch.empty(n_cam, n_rays, device=dev), torch.empty(n_cam, n_rays, device=dev)
        imgc, depc, mskc = torch.empty_like(img), torch.empty_like(dep), torch.empty_like(msk)
        ws = runtime.workspace(self, dev, L.holo_render_workspace_bytes(h, n_cam, 0))
        nul = C.c_void_p(None)
        opt = lambda t: runtime.ptr(t) if t is not None else nul  # noqa: E731
        _lib.check(L, L.holo_render_rays(h, runtime.ptr(grid), _camera_array(cams), n_cam, n_rays, runtime.ptr(xys), opt(u_c),
                                         opt(u_f), opt(nz_c), opt(nz_f), std, runtime.ptr(img), runtime.ptr(dep),
                                         runtime.ptr(msk), runtime.ptr(imgc), runtime.ptr(depc), runtime.ptr(mskc),
                                         runtime.ptr(ws), ws.numel(), runtime.stream_ptr(dev)), "holo_render_rays")
        if not as_output:
            return img, dep, msk, imgc, depc, mskc
        shp = lambda t, c: t.reshape(n_cam, c, n_rays, 1).permute(0, 2, 3, 1)  # noqa: E731  -> (n_cam, n_rays, 1, c)
        coarse = RendererOutput(features=shp(imgc, 3), depths=shp(depc, 1), masks=shp(mskc, 1))
        if not two_pass:
            return coarse
        return RendererOutput(features=shp(img, 3), depths=shp(dep, 1), masks=shp(msk, 1), prev_stage=coarse)

    def backward_training(self, bundle: ImplicitronRayBundle, implicit_functions, rng_streams: dict, grads: dict,
                          return_merged: bool = False):
        """Backward of the training-mode forward: what autograd computes in the reference for losses on
        the renderer's outputs (holo_diffusion_model.py:458-489).  ``rng_streams``: the draws of the forward call (all
        streams in use must be given).  ``grads``: gradients of the loss w.r.t. the forward's outputs, any subset of
        ``features (n_cam,n_rays,1,3)``, ``depths``, ``masks (n_cam,n_rays,1,1)`` and ``features_coarse`` / ``depths_coarse``
        / ``masks_coarse`` (the prev_stage outputs).  Returns ``(grad_grid (1,C,R,R,R), {RenderMLP parameter name: grad})``
        with the reference's state_dict names (``_density_net.mlp.<i>.0.weight`` ...).  The importance sampling carries no
        gradient (PyTorch3D's RayPointRefiner samples under torch.no_grad()).  ``return_merged``: also return the fine
        pass's depth list ``(n_cam, n_rays, P + Pf)`` and its is-importance-sample flags (the forward kernel's sample
        placement, held fixed by the backward pass)."""
        a = self._training_setup(bundle, implicit_functions, rng_streams, draw=False)
        h, grid, dev, cams, n_cam, n_rays = a["h"], a["grid"], a["dev"], a["cams"], a["n_cam"], a["n_rays"]
        if not a["two_pass"]:
            raise NotImplementedError("backward_training: the two-pass renderer (coarse + fine implicit function)")
        L = runtime.lib()
        nul = C.c_void_p(None)
        opt = lambda t: runtime.ptr(t) if t is not None else nul  # noqa: E731

        def g(key, c):
            t = grads.get(key)
            if t is None:
                return None
            if tuple(t.shape) != (n_cam, n_rays, 1, c):
                raise _lib.HoloError(f"grads['{key}'] must have shape {(n_cam, n_rays, 1, c)}, got {tuple(t.shape)}")
            return t.to(dev, torch.float32).permute(0, 3, 1, 2).reshape(n_cam, c, n_rays).contiguous()

        gi, gd, gm = g("features", 3), g("depths", 1), g("masks", 1)
        gic, gdc, gmc = g("features_coarse", 3), g("depths_coarse", 1), g("masks_coarse", 1)
        grad_grid = torch.empty_like(grid)
        zm = torch.empty(n_cam, n_rays, a["P"] + a["Pf"], device=dev) if return_merged else None
        zf = torch.empty(n_cam, n_rays, a["P"] + a["Pf"], device=dev, dtype=torch.uint8) if return_merged else None
        ws = runtime.workspace(self, dev, L.holo_render_rays_backward_workspace_bytes(h, n_cam, n_rays))
        _lib.check(L, L.holo_render_rays_backward(h, runtime.ptr(grid), _camera_array(cams), n_cam, n_rays, runtime.ptr(a["xys"]),
                                                  opt(a["u_c"]), opt(a["u_f"]), opt(a["nz_c"]), opt(a["nz_f"]), a["std"],
                                                  opt(gi), opt(gd), opt(gm), opt(gic), opt(gdc), opt(gmc), runtime.ptr(grad_grid),
                                                  opt(zm), opt(zf), runtime.ptr(ws), ws.numel(), runtime.stream_ptr(dev)),
                   "holo_render_rays_backward")
        pgrads = {}
        for name, prm in a["fn"].render_mlp.named_parameters():
            out = torch.empty_like(prm, dtype=torch.float32, device=dev)
            _lib.check(L, L.holo_renderer_get_grad(h, name.encode(), runtime.ptr(out), out.numel(), runtime.stream_ptr(dev)),
                       "holo_renderer_get_grad")
            pgrads[name] = out
        if return_merged:
            return grad_grid, pgrads, zm, zf
        return grad_grid, pgrads

    def forward(self, ray_bundle: ImplicitronRayBundle, implicit_functions: List[ImplicitFunctionWrapper],
                evaluation_mode: EvaluationMode = EvaluationMode.EVALUATION, rng_streams: Optional[dict] = None,
                **kwargs) -> RendererOutput:
        if evaluation_mode != EvaluationMode.EVALUATION:
            return self._forward_training(ray_bundle, implicit_functions, rng_streams)
        if not implicit_functions:
            raise ValueError("EA renderer expects implicit functions")
        wrapper = implicit_functions[0]
        fn = wrapper._fn
        if not isinstance(fn, HoloVoxelGridImplicitFunction):
            raise NotImplementedError("only HoloVoxelGridImplicitFunction is supported")
        if fn.feature_dim != 0 or fn.n_hidden not in (16, 32, 64):
            raise NotImplementedError("the fused renderer composites colours only and is built for 16/32/64 grid features "
                                      "(HoloDiffusionModel builds its implicit function with feature_dim = 0, "
                                      "holo_diffusion_model.py:156); feature_dim > 0 / n_hidden = 128 are served by the "
                                      "stand-alone HoloVoxelGridImplicitFunction.forward")
        grid = wrapper.bound_args.get("voxel_grid_features")
        if grid is None:
            raise ValueError("voxel_grid_features must be bound to the implicit function (bind_args)")
        runtime.require_device(grid, "HoloMultiPassEmissionAbsorptionRenderer.forward")
        dev = grid.device
        if tuple(grid.shape) != (1, fn.n_hidden, fn.resol, fn.resol, fn.resol):
            raise _lib.HoloError(f"voxel grid must be (1,{fn.n_hidden},{fn.resol}^3), got {tuple(grid.shape)}")
        h = self._ensure_handle(fn, ray_bundle, dev)
        L = runtime.lib()
        cams = ray_bundle.camera
        n_cam = len(cams)
        H, W = ray_bundle.image_height, ray_bundle.image_width
        arr = _camera_array(cams)
        grid = grid.contiguous().float()
        img = torch.empty(n_cam, 3, H, W, device=dev)
        dep = torch.empty(n_cam, 1, H, W, device=dev)
        msk = torch.empty(n_cam, 1, H, W, device=dev)
        imgc, depc, mskc = torch.empty_like(img), torch.empty_like(dep), torch.empty_like(msk)
        # render_normals (released YAMLs set it on the implicit function, configs/apple.yaml:203): the normals of both
        # passes are composited inside the same kernel (holo_multipass_ea.py:105-109)
        want_normals = bool(getattr(fn, "render_normals", False))
        nrm = torch.empty_like(img) if want_normals else None
        nrmc = torch.empty_like(img) if want_normals else None
        nbytes = L.holo_render_workspace_bytes(h, n_cam, 1 if want_normals else 0)
        ws = runtime.workspace(self, dev, nbytes)
        nul = C.c_void_p(None)
        _lib.check(L, L.holo_render(h, runtime.ptr(grid), arr, n_cam, runtime.ptr(img), runtime.ptr(dep),
                                    runtime.ptr(msk), runtime.ptr(imgc), runtime.ptr(depc), runtime.ptr(mskc),
                                    runtime.ptr(nrm) if want_normals else nul, runtime.ptr(nrmc) if want_normals else nul,
                                    runtime.ptr(ws), ws.numel(), runtime.stream_ptr(dev)), "holo_render")
        coarse = RendererOutput(features=imgc.permute(0, 2, 3, 1), depths=depc.permute(0, 2, 3, 1),
                                masks=mskc.permute(0, 2, 3, 1),
                                normals=nrmc.permute(0, 2, 3, 1) if want_normals else None)
        if len(implicit_functions) == 1:
            return coarse
        return RendererOutput(features=img.permute(0, 2, 3, 1), depths=dep.permute(0, 2, 3, 1),
                              masks=msk.permute(0, 2, 3, 1), prev_stage=coarse,
                              normals=nrm.permute(0, 2, 3, 1) if want_normals else None)


class _HoloRenderRaysFn(torch.autograd.Function):
    """Autograd node of the training-mode renderer: forward = holo_render_rays, backward = holo_render_rays_backward with
    the SAME ray list and random draws.  Inputs that can receive a gradient: the voxel grid and the RenderMLP parameters
    (``names`` gives their order); outputs: the six raw planes rgb (n_cam,3,n_rays), depth, mask of the fine and the coarse
    pass."""

    @staticmethod
    def forward(ctx, renderer, bundle, implicit_functions, streams, names, grid, *params):
        ctx.renderer, ctx.bundle, ctx.fns, ctx.streams, ctx.names = renderer, bundle, list(implicit_functions), streams, names
        ctx.grid_needs = grid.requires_grad
        ctx.save_for_backward(grid.detach())
        with torch.no_grad():
            a = renderer._training_setup(bundle, implicit_functions, streams, draw=False)
            return renderer._render_rays_raw(a, as_output=False)

    @staticmethod
    def backward(ctx, g_img, g_dep, g_msk, g_imgc, g_depc, g_mskc):
        (grid,) = ctx.saved_tensors
        n_cam, n_rays = g_img.shape[0], g_img.shape[2]
        to4 = lambda t, c: None if t is None else t.reshape(n_cam, c, n_rays, 1).permute(0, 2, 3, 1).contiguous()  # noqa: E731
        grads = {k: v for k, v in (("features", to4(g_img, 3)), ("depths", to4(g_dep, 1)), ("masks", to4(g_msk, 1)),
                                   ("features_coarse", to4(g_imgc, 3)), ("depths_coarse", to4(g_depc, 1)),
                                   ("masks_coarse", to4(g_mskc, 1))) if v is not None}
        # the grid of the forward call is bound again for the backward kernels (the caller may have unbound it since)
        saved = [dict(f.bound_args) for f in ctx.fns]
        for f in ctx.fns:
            f.bind_args(voxel_grid_features=grid)
        try:
            with torch.no_grad():
                g_grid, pg = ctx.renderer.backward_training(ctx.bundle, ctx.fns, ctx.streams, grads)
        finally:
            for f, b in zip(ctx.fns, saved):
                f.unbind_args()
                if b:
                    f.bind_args(**b)
        wanted = ctx.needs_input_grad[6:]
        return (None, None, None, None, None, g_grid if ctx.grid_needs else None) + \
            tuple(pg[k] if need else None for k, need in zip(ctx.names, wanted))
