"""Encoder-side plugins: source-view feature maps -> voxel feature grid on the HIP path.

Reference interfaces mirrored (paths relative to /root/reference/holo_diffusion):
  * the ``view_pooler`` member of the model (PyTorch3D ``ViewPooler`` = ``ViewSampler`` + feature aggregator), called
    with ``pts = VolumeLocator.get_coord_grid()`` at holo_diffusion_model.py:349-367; released configuration
    configs/apple.yaml:183-196 (``masked_sampling: false``, ``sampling_mode: bilinear``,
    ``AngleWeightedReductionFeatureAggregator`` with ``[AVG, STD]``, gamma 1.0, min weight 0.1); the model forces
    ``exclude_target_view = exclude_target_view_mask_features = False`` (:114-116)
  * ``pooled_feature_mapper`` (``LazyLinearWithXavierInit(feature_size)``, :113) and ``tanh`` (:368-373)

All arithmetic - projection, bilinear gather, angle-weighted AVG/STD reduction, the mapper's Linear and the tanh - runs
in ONE kernel (``holo_view_pool``, csrc/kernels_viewpool.hip); the classes here carry the configuration.  The image
feature extractor (PyTorch3D's ResNetFeatureExtractor) is outside this path: its output dict is the input here.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Dict, Optional, Tuple

import torch

from . import _lib, runtime
from .registry import Configurable, apply_config, pt3d_base, registry


class ReductionFunction(enum.Enum):
    AVG = "avg"
    MAX = "max"
    STD = "std"
    STD_AVG = "std_avg"


class ViewSampler(Configurable):
    masked_sampling: bool = False
    sampling_mode: str = "bilinear"

    def __init__(self, **kwargs):
        apply_config(self, kwargs)
        if self.masked_sampling or self.sampling_mode != "bilinear":
            raise NotImplementedError("ViewSampler: the fused kernel supports the released configuration "
                                      "(masked_sampling false, bilinear; configs/apple.yaml:185-187)")


FeatureAggregatorBase = pt3d_base("implicitron.models.view_pooler.feature_aggregator", "FeatureAggregatorBase")


@registry.register_local  # never into PyTorch3D's registry: it would silently replace PyTorch3D's own aggregator
class AngleWeightedReductionFeatureAggregator(FeatureAggregatorBase):
    exclude_target_view: bool = True
    exclude_target_view_mask_features: bool = True
    concatenate_output: bool = True
    reduction_functions: Tuple[ReductionFunction, ...] = (ReductionFunction.AVG, ReductionFunction.STD)
    weight_by_ray_angle_gamma: float = 1.0
    min_ray_angle_weight: float = 0.1

    def __init__(self, **kwargs):
        apply_config(self, kwargs)
        self.reduction_functions = tuple(ReductionFunction[r] if isinstance(r, str) else r
                                         for r in self.reduction_functions)
        if self.reduction_functions != (ReductionFunction.AVG, ReductionFunction.STD) or not self.concatenate_output:
            raise NotImplementedError("AngleWeightedReductionFeatureAggregator: the fused kernel implements the released "
                                      "reduction [AVG, STD] with concatenated output (configs/apple.yaml:188-196)")

    def get_aggregated_feature_dim(self, feats_or_feats_dim) -> int:
        d = feats_or_feats_dim if isinstance(feats_or_feats_dim, int) else sum(
            int(t.shape[1]) for t in feats_or_feats_dim.values())
        return len(self.reduction_functions) * d


@registry.register_local  # (the reference registers ITS class under this name, custom_modules.py:162-163: never overwrite it)
class MLPMeanFeatureAggregator(torch.nn.Module, FeatureAggregatorBase):
    """The learnt aggregator of configs/hydrant.yaml:184 / old_base_config.yaml:205 (custom_modules.py:162-293): config
    fields :171-176 (+ PyTorch3D's three FeatureAggregatorBase fields), parameters under the reference's names
    ``_first_sampled`` / ``_first_mean`` (LazyLinear: sized by the first checkpoint or pooled batch), ``_mlp.mlp.0.0``,
    ``_last``.  The arithmetic runs in ``holo_mlp_mean_pool`` (csrc/kernels_viewpool.hip), fused with the sampling in
    front of it and the model's ``pooled_feature_mapper`` + tanh behind it."""
    exclude_target_view: bool = True
    exclude_target_view_mask_features: bool = True
    concatenate_output: bool = True
    n_hidden: int = 128
    dim_out: int = 128
    n_layers: int = 1
    n_harmonic_functions_ray: int = 3
    checkpointed_mlp: bool = True

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        self.__dict__["_native"] = None  # [handle, key, parameter versions] of the folded native copy
        apply_config(self, kwargs)
        if self.n_hidden != 128 or self.n_layers != 1 or not self.concatenate_output:
            raise NotImplementedError("MLPMeanFeatureAggregator: the fused kernel implements the released configuration "
                                      "(n_hidden 128, n_layers 1, concatenated output; configs/hydrant.yaml:188-196)")
        nh = self.n_hidden
        self._first_sampled = torch.nn.LazyLinear(nh)  # LazyLinearWithXavierInit (custom_modules.py:36-41)
        self._first_mean = torch.nn.LazyLinear(nh)
        self._last = torch.nn.Linear(nh, self.dim_out)
        torch.nn.init.xavier_uniform_(self._last.weight)
        self._mlp = torch.nn.Module()
        self._mlp.mlp = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(nh, nh))])  # names `_mlp.mlp.0.0.*`
        torch.nn.init.xavier_uniform_(self._mlp.mlp[0][0].weight)
        for p in self.parameters():
            if not isinstance(p, torch.nn.parameter.UninitializedParameter):
                p.requires_grad_(False)

    def get_aggregated_feature_dim(self, feats_or_feats_dim) -> int:
        return self.dim_out

    def input_dim(self, feats: Dict[str, torch.Tensor]) -> int:
        return sum(int(t.shape[1]) for t in feats.values()) + 3 * (2 * self.n_harmonic_functions_ray + 1)

    def materialize(self, in_dim: int, device) -> None:
        """First use of the two LazyLinear layers (Xavier weights, zero bias: custom_modules.py:36-41)."""
        for lin in (self._first_sampled, self._first_mean):
            if isinstance(lin.weight, torch.nn.parameter.UninitializedParameter):
                lin.in_features = in_dim
                lin.weight.materialize((self.n_hidden, in_dim), device=device)
                lin.bias.materialize((self.n_hidden,), device=device)
                torch.nn.init.xavier_uniform_(lin.weight.data)
                lin.bias.data.zero_()
                lin.__class__ = torch.nn.Linear
                lin.weight.requires_grad_(False)
                lin.bias.requires_grad_(False)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("MLPMeanFeatureAggregator: aggregation at arbitrary points is not on this path; the model "
                                  "pools onto its voxel grid through ViewPooler.pool_to_voxel_features() (one fused kernel)")

    def close(self):
        nat = self.__dict__.get("_native")
        if nat is not None:
            try:
                runtime.lib().holo_mlp_mean_destroy(nat[0])
            except Exception:
                pass
            self.__dict__["_native"] = None

    def __del__(self):
        self.close()


class ViewPooler(Configurable, torch.nn.Module):
    view_sampler_args: Optional[dict] = None
    feature_aggregator_class_type: str = "AngleWeightedReductionFeatureAggregator"
    feature_aggregator_AngleWeightedReductionFeatureAggregator_args: Optional[dict] = None
    feature_aggregator_MLPMeanFeatureAggregator_args: Optional[dict] = None

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        apply_config(self, kwargs)
        self.view_sampler = ViewSampler(**dict(self.view_sampler_args or {}))
        agg_type = registry.get(FeatureAggregatorBase, self.feature_aggregator_class_type)
        self.feature_aggregator = agg_type(**dict(
            getattr(self, f"feature_aggregator_{self.feature_aggregator_class_type}_args", None) or {}))

    def get_aggregated_feature_dim(self, feats) -> int:
        return self.feature_aggregator.get_aggregated_feature_dim(feats)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("ViewPooler: pooling at arbitrary points is not on this path; the model pools onto its "
                                  "voxel grid through pool_to_voxel_features() (one fused kernel incl. the mapper)")

    @torch.no_grad()
    def pool_to_voxel_features(self, feats: Dict[str, torch.Tensor], camera, mapper_weight: torch.Tensor,
                               mapper_bias: Optional[torch.Tensor], resol: int, volume_extent: float) -> torch.Tensor:
        """feats: key -> (n_src, C_k, H_k, W_k) float32 on the device (the image feature extractor's dict, in its
        order); camera: the n_src source cameras.  Returns tanh(mapper(aggregated)) as (1, F, R, R, R)."""
        agg = self.feature_aggregator
        if agg.exclude_target_view or agg.exclude_target_view_mask_features:
            raise _lib.HoloError("view pooling: exclude_target_view(_mask_features) must be False "
                                 "(HoloDiffusionModel sets both, holo_diffusion_model.py:114-116)")
        if isinstance(agg, MLPMeanFeatureAggregator):
            return self._pool_mlp_mean(feats, camera, mapper_weight, mapper_bias, resol, volume_extent)
        from .render import _camera_array
        keys = list(feats)
        if not keys:
            raise ValueError("view pooling needs at least one feature map")
        t0 = feats[keys[0]]
        runtime.require_device(t0, "ViewPooler.pool_to_voxel_features")
        dev, n_src = t0.device, int(t0.shape[0])
        arr = (_lib.HoloViewFeature * len(keys))()
        held = []
        for i, k in enumerate(keys):
            t = feats[k]
            if t.dim() != 4 or t.shape[0] != n_src or t.device != dev:
                raise _lib.HoloError(f"feature map '{k}' must be (n_src={n_src}, C, H, W) on {dev}, got {tuple(t.shape)}")
            t = t.contiguous().float()
            held.append(t)
            arr[i].feats = t.data_ptr()
            arr[i].channels, arr[i].height, arr[i].width = int(t.shape[1]), int(t.shape[2]), int(t.shape[3])
        cams = _camera_array(camera)
        if len(cams) != n_src:
            raise _lib.HoloError(f"{len(cams)} cameras for {n_src} source views")
        F = int(mapper_weight.shape[0])
        A = self.get_aggregated_feature_dim({k: feats[k] for k in keys})
        if tuple(mapper_weight.shape) != (F, A):
            raise _lib.HoloError(f"pooled_feature_mapper.weight must be ({F}, {A}), got {tuple(mapper_weight.shape)}")
        cfg = _lib.HoloViewPoolCfg(int(resol), float(volume_extent), F, float(agg.weight_by_ray_angle_gamma),
                                   float(agg.min_ray_angle_weight), 1e-2)
        L = runtime.lib()
        nbytes = L.holo_view_pool_workspace_bytes(C.byref(cfg), arr, len(keys), n_src)
        ws = runtime.workspace(self, dev, nbytes)
        out = torch.empty(1, F, resol, resol, resol, device=dev)
        w = mapper_weight.detach().contiguous().float()
        b = mapper_bias.detach().contiguous().float() if mapper_bias is not None else None
        _lib.check(L, L.holo_view_pool(runtime.ctx(dev), C.byref(cfg), arr, len(keys), cams, n_src, runtime.ptr(w),
                                       runtime.ptr(b) if b is not None else C.c_void_p(None), runtime.ptr(out),
                                       runtime.ptr(ws), ws.numel(), runtime.stream_ptr(dev)), "holo_view_pool")
        return out

    @torch.no_grad()
    def pool_to_voxel_features_backward(self, feats: Dict[str, torch.Tensor], camera, mapper_weight: torch.Tensor,
                                        mapper_bias: Optional[torch.Tensor], resol: int, volume_extent: float,
                                        grad_voxel_features: torch.Tensor, want_feature_grads: bool = True):
        """Backward of ``pool_to_voxel_features`` (``holo_view_pool_backward``): the gradient of the clean grid
        (``training_backward(...)["voxel_features"]``) -> ``({key: grad of feats[key]}, grad mapper weight, grad mapper
        bias)`` - what autograd leaves behind holo_diffusion_model.py:358-373 for the image feature extractor and on
        ``pooled_feature_mapper``.  AngleWeightedReductionFeatureAggregator (the released configuration) only."""
        agg = self.feature_aggregator
        if isinstance(agg, MLPMeanFeatureAggregator):
            return self._pool_mlp_mean_backward(feats, camera, mapper_weight, mapper_bias, resol, volume_extent,
                                                grad_voxel_features, want_feature_grads)
        if agg.exclude_target_view or agg.exclude_target_view_mask_features:
            raise _lib.HoloError("view pooling: exclude_target_view(_mask_features) must be False "
                                 "(HoloDiffusionModel sets both, holo_diffusion_model.py:114-116)")
        from .render import _camera_array
        keys = list(feats)
        t0 = feats[keys[0]]
        runtime.require_device(t0, "ViewPooler.pool_to_voxel_features_backward")
        dev, n_src = t0.device, int(t0.shape[0])
        arr = (_lib.HoloViewFeature * len(keys))()
        held, grads = [], {}
        gptr = (C.c_void_p * len(keys))()
        for i, k in enumerate(keys):
            t = feats[k]
            if t.dim() != 4 or t.shape[0] != n_src or t.device != dev:
                raise _lib.HoloError(f"feature map '{k}' must be (n_src={n_src}, C, H, W) on {dev}, got {tuple(t.shape)}")
            t = t.detach().contiguous().float()
            held.append(t)
            arr[i].feats = t.data_ptr()
            arr[i].channels, arr[i].height, arr[i].width = int(t.shape[1]), int(t.shape[2]), int(t.shape[3])
            if want_feature_grads:
                grads[k] = torch.empty_like(t)
                gptr[i] = grads[k].data_ptr()
            else:
                gptr[i] = None
        cams = _camera_array(camera)
        if len(cams) != n_src:
            raise _lib.HoloError(f"{len(cams)} cameras for {n_src} source views")
        F = int(mapper_weight.shape[0])
        A = self.get_aggregated_feature_dim({k: feats[k] for k in keys})
        if tuple(mapper_weight.shape) != (F, A):
            raise _lib.HoloError(f"pooled_feature_mapper.weight must be ({F}, {A}), got {tuple(mapper_weight.shape)}")
        g = grad_voxel_features.detach().contiguous().float()
        if tuple(g.shape) != (1, F, resol, resol, resol) or g.device != dev:
            raise _lib.HoloError(f"grad_voxel_features must be (1, {F}, {resol}, {resol}, {resol}) on {dev}, got {tuple(g.shape)}")
        cfg = _lib.HoloViewPoolCfg(int(resol), float(volume_extent), F, float(agg.weight_by_ray_angle_gamma),
                                   float(agg.min_ray_angle_weight), 1e-2)
        L = runtime.lib()
        ctx = runtime.ctx(dev)
        nbytes = L.holo_view_pool_backward_workspace_bytes(ctx, C.byref(cfg), arr, len(keys), n_src)
        ws = runtime.workspace(self, dev, nbytes)
        w = mapper_weight.detach().contiguous().float()
        b = mapper_bias.detach().contiguous().float() if mapper_bias is not None else None
        gw = torch.empty_like(w)
        gb = torch.empty(F, device=dev)
        _lib.check(L, L.holo_view_pool_backward(ctx, C.byref(cfg), arr, len(keys), cams, n_src, runtime.ptr(w),
                                                runtime.ptr(b) if b is not None else C.c_void_p(None), runtime.ptr(g), gptr,
                                                runtime.ptr(gw), runtime.ptr(gb), runtime.ptr(ws), ws.numel(),
                                                runtime.stream_ptr(dev)), "holo_view_pool_backward")
        return grads, gw, (gb if mapper_bias is not None else None)

    def _mlp_mean_native(self, feats: Dict[str, torch.Tensor], camera, mapper_weight, mapper_bias, resol: int,
                         volume_extent: float):
        """The native pooler of the MLPMean aggregator with the current parameters committed, and the call's arrays."""
        from .render import _camera_array
        agg: MLPMeanFeatureAggregator = self.feature_aggregator
        keys = list(feats)
        t0 = feats[keys[0]]
        runtime.require_device(t0, "ViewPooler.pool_to_voxel_features")
        dev, n_src = t0.device, int(t0.shape[0])
        agg.materialize(agg.input_dim(feats), dev)
        F = int(mapper_weight.shape[0])
        if tuple(mapper_weight.shape) != (F, agg.dim_out):
            raise _lib.HoloError(f"pooled_feature_mapper.weight must be ({F}, {agg.dim_out}), got {tuple(mapper_weight.shape)}")
        arr = (_lib.HoloViewFeature * len(keys))()
        held = []
        for i, k in enumerate(keys):
            t = feats[k]
            if t.dim() != 4 or t.shape[0] != n_src or t.device != dev:
                raise _lib.HoloError(f"feature map '{k}' must be (n_src={n_src}, C, H, W) on {dev}, got {tuple(t.shape)}")
            t = t.contiguous().float()
            held.append(t)
            arr[i].feats = t.data_ptr()
            arr[i].channels, arr[i].height, arr[i].width = int(t.shape[1]), int(t.shape[2]), int(t.shape[3])
        cams = _camera_array(camera)
        if len(cams) != n_src:
            raise _lib.HoloError(f"{len(cams)} cameras for {n_src} source views")
        L = runtime.lib()
        params = {k: p for k, p in agg.named_parameters()}
        params["pooled_feature_mapper.weight"] = mapper_weight
        if mapper_bias is None:  # one cached zero row (a fresh tensor per call would change the fingerprint below every time)
            zb = agg.__dict__.get("_zero_mapper_bias")
            if zb is None or zb.device != dev or zb.numel() != F:
                zb = torch.zeros(F, device=dev)
                agg.__dict__["_zero_mapper_bias"] = zb
            mapper_bias_t = zb
        else:
            mapper_bias_t = mapper_bias
        params["pooled_feature_mapper.bias"] = mapper_bias_t
        key = (dev, int(resol), float(volume_extent), F, tuple(int(a.channels) for a in arr))
        versions = tuple((k, p.data_ptr(), p._version) for k, p in params.items())
        if agg._native is None or agg._native[1] != key:
            agg.close()
            cfg = _lib.HoloMlpMeanCfg()
            cfg.resol, cfg.volume_extent, cfg.feature_size = int(resol), float(volume_extent), F
            cfg.n_hidden, cfg.dim_out, cfg.n_layers = int(agg.n_hidden), int(agg.dim_out), int(agg.n_layers)
            cfg.n_harmonic_functions_ray, cfg.n_feats, cfg.projection_eps = int(agg.n_harmonic_functions_ray), len(keys), 1e-2
            for i in range(len(keys)):
                cfg.channels[i] = int(arr[i].channels)
            h = C.c_void_p()
            _lib.check(L, L.holo_mlp_mean_create(runtime.ctx(dev), C.byref(cfg), C.byref(h)), "holo_mlp_mean_create")
            agg.__dict__["_native"] = [h, key, None]
        h = agg._native[0]
        if agg._native[2] != versions:
            st = runtime.stream_ptr(dev)
            for k, p in params.items():
                if p.device != dev or p.dtype != torch.float32:
                    raise _lib.HoloError(f"aggregator parameter '{k}' is {p.dtype} on {p.device}; expected float32 on {dev}")
                t = p.detach().contiguous()
                _lib.check(L, L.holo_mlp_mean_set_param(h, k.encode(), runtime.ptr(t), t.dim(), _lib.shape_array(t.shape), st),
                           f"holo_mlp_mean_set_param({k})")
            _lib.check(L, L.holo_mlp_mean_commit(h, st), "holo_mlp_mean_commit")
            agg._native[2] = versions
        return h, arr, held, cams, n_src, dev, F, keys, params

    @torch.no_grad()
    def _pool_mlp_mean_backward(self, feats, camera, mapper_weight, mapper_bias, resol, volume_extent, grad_voxel_features,
                                want_feature_grads):
        """``holo_mlp_mean_backward``: as ``pool_to_voxel_features_backward``; the aggregator's own parameter gradients are
        left in ``self.feature_aggregator.native_grads`` (reference names ``_first_sampled.weight`` ...)."""
        h, arr, held, cams, n_src, dev, F, keys, params = self._mlp_mean_native(feats, camera, mapper_weight, mapper_bias, resol,
                                                                              volume_extent)
        g = grad_voxel_features.detach().contiguous().float()
        if tuple(g.shape) != (1, F, resol, resol, resol) or g.device != dev:
            raise _lib.HoloError(f"grad_voxel_features must be (1, {F}, {resol}, {resol}, {resol}) on {dev}, got {tuple(g.shape)}")
        grads = {}
        gptr = (C.c_void_p * len(keys))()
        for i, k in enumerate(keys):
            if want_feature_grads:
                grads[k] = torch.empty_like(held[i])
                gptr[i] = grads[k].data_ptr()
            else:
                gptr[i] = None
        L = runtime.lib()
        ws = runtime.workspace(self, dev, L.holo_mlp_mean_backward_workspace_bytes(h, arr, len(keys), n_src))
        st = runtime.stream_ptr(dev)
        _lib.check(L, L.holo_mlp_mean_backward(h, arr, len(keys), cams, n_src, runtime.ptr(g), gptr, runtime.ptr(ws), ws.numel(),
                                               st), "holo_mlp_mean_backward")
        pg = {}
        for k, p in params.items():
            t = torch.empty(p.shape, device=dev, dtype=torch.float32)
            _lib.check(L, L.holo_mlp_mean_get_grad(h, k.encode(), runtime.ptr(t), t.numel(), st), f"holo_mlp_mean_get_grad({k})")
            pg[k] = t
        gw = pg.pop("pooled_feature_mapper.weight")
        gb = pg.pop("pooled_feature_mapper.bias")
        self.feature_aggregator.__dict__["native_grads"] = pg
        return grads, gw, (gb if mapper_bias is not None else None)

    @torch.no_grad()
    def _pool_mlp_mean(self, feats: Dict[str, torch.Tensor], camera, mapper_weight, mapper_bias, resol: int,
                       volume_extent: float) -> torch.Tensor:
        h, arr, held, cams, n_src, dev, F, keys, _ = self._mlp_mean_native(feats, camera, mapper_weight, mapper_bias, resol,
                                                                         volume_extent)
        L = runtime.lib()
        ws = runtime.workspace(self, dev, L.holo_mlp_mean_workspace_bytes(h, arr, len(keys), n_src))
        out = torch.empty(1, F, resol, resol, resol, device=dev)
        _lib.check(L, L.holo_mlp_mean_pool(h, arr, len(keys), cams, n_src, runtime.ptr(out), runtime.ptr(ws), ws.numel(),
                                           runtime.stream_ptr(dev)), "holo_mlp_mean_pool")
        return out
