"""``HoloDiffusionModel`` — the model plugin the sampling scripts drive.

Mirrors /root/reference/holo_diffusion/holo_diffusion_model.py:
  * config fields resol / volume_extent / feature_size / num_passes / net_3d_* / diffusion_* (:47-61)
    plus the GenericModel fields the released YAMLs set (configs/apple.yaml:104-165)
  * ``create_net_3d`` (:118-130): net_3d args overridden with in/out_channels=feature_size, image_size=resol
  * ``create_diffusion`` (:132-136)
  * ``_construct_implicit_functions`` (:138-171): resol / volume_extent / n_hidden=feature_size /
    feature_dim=0 forced, ONE implicit function shared by all passes
  * ``sample_random_voxel_features`` (:188-199) and ``..._progressive`` (:173-186)
  * ``forward`` EVALUATION branch with ``voxel_features`` (:247-326,376-540): range checks,
    ``tanh(net_3d(vf, t=0))``, bind, ray sampler, render, ``images_render / depths_render / masks_render``

Training-side branches (view pooling :327-374, diffusion training/bootstrap :386-418, losses) are
outside the hot path ("next" rows).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _lib, runtime
from .cameras import PerspectiveCameras
from .diffusion import ImplicitronGaussianDiffusion
from .registry import apply_config, get_default_args, pt3d_base, registry
from .render import (AdaptiveRaySampler, BaseRenderer, EvaluationMode, HoloMultiPassEmissionAbsorptionRenderer,
                     HoloVoxelGridImplicitFunction, ImplicitFunctionBase, ImplicitFunctionWrapper, RenderSamplingMode)
from .unet import SimpleUnet3D, Unet3DBase

logger = logging.getLogger(__name__)


# pytorch3d.implicitron.models.base_model.ImplicitronModelBase when PyTorch3D is importable (so that Implicitron's
# ModelFactory resolves `model_class_type: HoloDiffusionModel` to the class below), a stand-in otherwise
ImplicitronModelBase = pt3d_base("implicitron.models.base_model", "ImplicitronModelBase")


@registry.register
class HoloDiffusionModel(ImplicitronModelBase, torch.nn.Module):
    # ---- model config (holo_diffusion_model.py:47-61)
    resol: int = 32
    volume_extent: float = 8.0
    feature_size: int = 128
    num_passes: int = 2
    net_3d_enabled: bool = True
    net_3d_class_type: str = "SimpleUnet3D"
    net_3d_SimpleUnet3D_args: Optional[dict] = None
    diffusion_enabled: bool = True
    diffusion_args: Optional[dict] = None
    enable_bootstrap: bool = True
    bootstrap_prob: float = 0.5
    # ---- GenericModel fields used on the path (configs/apple.yaml:104-165)
    render_image_width: int = 400
    render_image_height: int = 400
    bg_color: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    chunk_size_grid: int = 163840        # accepted for config compatibility; the fused renderer is chunk-free
    n_train_target_views: int = 10
    sampling_mode_training: str = "mask_sample"
    sampling_mode_evaluation: str = "full_grid"
    raysampler_class_type: str = "AdaptiveRaySampler"
    raysampler_AdaptiveRaySampler_args: Optional[dict] = None
    renderer_class_type: str = "HoloMultiPassEmissionAbsorptionRenderer"
    renderer_HoloMultiPassEmissionAbsorptionRenderer_args: Optional[dict] = None
    implicit_function_class_type: str = "HoloVoxelGridImplicitFunction"
    implicit_function_HoloVoxelGridImplicitFunction_args: Optional[dict] = None
    # ---- encoder side (holo_diffusion_model.py:113-116,327-374).  Off by default HERE (the released
    # YAMLs enable it): the sampling drivers never pool views; with it on the model owns `pooled_feature_mapper`.
    view_pooler_enabled: bool = False
    view_pooler_args: Optional[dict] = None
    # host-side switch: the reference re-runs tanh(net_3d(vf, 0)) for EVERY rendered frame even when vf is
    # unchanged (holo_diffusion_model.py:420-426); cache it on the identity of the voxel_features tensor.
    cache_refined_features: bool = True
    check_ranges: bool = False           # the reference's per-call min/max asserts (:381,426,428) force host syncs

    def __init__(self, **kwargs):
        torch.nn.Module.__init__(self)
        apply_config(self, kwargs)
        self.create_net_3d()
        self.create_diffusion()
        rs_args = dict(self.raysampler_AdaptiveRaySampler_args or {})
        rs_args.setdefault("scene_extent", 4.0)
        rs_args["image_width"], rs_args["image_height"] = self.render_image_width, self.render_image_height
        if self.raysampler_class_type != "AdaptiveRaySampler":
            raise NotImplementedError("only AdaptiveRaySampler is supported")
        self.raysampler = AdaptiveRaySampler(**rs_args)
        r_args = dict(self.renderer_HoloMultiPassEmissionAbsorptionRenderer_args or {})
        rm = dict(r_args.get("raymarcher_EmissionAbsorptionRaymarcher_args") or {})
        rm.setdefault("bg_color", tuple(self.bg_color))  # GenericModel passes its bg_color to the raymarcher
        r_args["raymarcher_EmissionAbsorptionRaymarcher_args"] = rm
        self.renderer = registry.get(BaseRenderer, self.renderer_class_type)(**r_args)
        self._implicit_functions = self._construct_implicit_functions()
        self._refined_cache = None
        self.view_pooler = None
        self.image_feature_extractor = None  # any callable (image_rgb, fg_probability) -> {key: (n, C, H, W)}; the
        # reference's ResNetFeatureExtractor is outside this path
        if self.view_pooler_enabled:
            from .viewpool import ViewPooler
            self.view_pooler = ViewPooler(**dict(self.view_pooler_args or {}))
            # (holo_diffusion_model.py:114-116: "Setting target view exclusion to False by hard!")
            self.view_pooler.feature_aggregator.exclude_target_view = False
            self.view_pooler.feature_aggregator.exclude_target_view_mask_features = False
            self.pooled_feature_mapper = torch.nn.LazyLinear(self.feature_size)  # LazyLinearWithXavierInit (:113)

    def create_net_3d(self):
        self.net_3d = None
        if self.net_3d_enabled:
            extra = dict(in_channels=self.feature_size, out_channels=self.feature_size, image_size=self.resol)
            args = dict(getattr(self, "net_3d_" + self.net_3d_class_type + "_args") or {})
            self.net_3d = registry.get(Unet3DBase, self.net_3d_class_type)(**{**args, **extra})

    def create_diffusion(self):
        self.diffusion = None
        if self.diffusion_enabled:
            self.diffusion = ImplicitronGaussianDiffusion(**dict(self.diffusion_args or {}))

    def _construct_implicit_functions(self):
        if self.implicit_function_class_type != "HoloVoxelGridImplicitFunction":
            raise ValueError(f"{type(self)} supports only HoloVoxelGridImplicitFunction!")
        extra = dict(resol=self.resol, volume_extent=self.volume_extent, n_hidden=self.feature_size, feature_dim=0)
        cfg = dict(self.implicit_function_HoloVoxelGridImplicitFunction_args or {})
        fn_type = registry.get(ImplicitFunctionBase, self.implicit_function_class_type)
        if_ = ImplicitFunctionWrapper(fn_type(**{**cfg, **extra}))
        return torch.nn.ModuleList([if_ for _ in range(self.num_passes)])

    # ---- sampling (holo_diffusion_model.py:173-199) -----------------------------------------
    def _shape(self):
        return (1, self.feature_size, self.resol, self.resol, self.resol)

    def sample_random_voxel_features_progressive(self, **loop_kwargs):
        assert self.net_3d_enabled and self.diffusion_enabled
        for sample in self.diffusion.p_sample_loop_progressive(model=self.net_3d, shape=self._shape(),
                                                               clip_denoised=True, progress=False, **loop_kwargs):
            s = sample["sample"]
            out = torch.empty_like(s)
            L = runtime.lib()
            _lib.check(L, L.holo_clip(runtime.ctx(s.device), runtime.ptr(s), runtime.ptr(out), -1.0, 1.0, s.numel(),
                                      runtime.stream_ptr(s.device)), "holo_clip")
            yield out

    def sample_random_voxel_features(self, **loop_kwargs) -> torch.Tensor:
        assert self.net_3d_enabled and self.diffusion_enabled
        logger.info("generating random voxel features through denoising diffusion ...")
        return self.diffusion.p_sample_loop(model=self.net_3d, shape=self._shape(), clip_denoised=True,
                                            progress=loop_kwargs.pop("progress", False), **loop_kwargs)

    # ---- render (holo_diffusion_model.py:201-540, evaluation branch) ------------------------
    def _weights_epoch(self):
        net = self.net_3d
        epoch = net.weights_epoch() if hasattr(net, "weights_epoch") else getattr(net, "_weights_epoch", 0)
        return (epoch, getattr(net, "compute_dtype", None), id(net))

    def invalidate_refined_cache(self) -> None:
        self._refined_cache = None

    def _apply(self, fn, *a, **k):
        self._refined_cache = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._refined_cache = None
        return super().load_state_dict(*a, **k)

    def _refine(self, voxel_features: torch.Tensor) -> torch.Tensor:
        """tanh(net_3d(vf, t=0)) (:420-426).

        The cache HOLDS the input tensor: while it is cached its storage cannot be handed to another tensor by the
        caching allocator, so identity (``is``) + ``_version`` + the denoiser's weight epoch / compute mode identify
        the input.  (Grids written through raw pointers by the library are always fresh tensors: holo_ddpm_step and
        holo_clip never write in place.)"""
        c = self._refined_cache
        if (self.cache_refined_features and c is not None and c[0] is voxel_features
                and c[1] == voxel_features._version and c[2] == self._weights_epoch()):
            return c[3]
        dev = voxel_features.device
        t0 = torch.zeros((1,), dtype=torch.long, device=dev)
        y = self.net_3d(voxel_features, t0)
        out = torch.empty_like(y)
        L = runtime.lib()
        _lib.check(L, L.holo_tanh(runtime.ctx(dev), runtime.ptr(y), runtime.ptr(out), y.numel(),
                                  runtime.stream_ptr(dev)), "holo_tanh")
        if self.cache_refined_features:
            self._refined_cache = (voxel_features, voxel_features._version, self._weights_epoch(), out)
        else:
            self._refined_cache = None
        return out

    def forward(self, *, image_rgb=None, camera: PerspectiveCameras, fg_probability=None, mask_crop=None,
                depth_map=None, sequence_name=None, frame_timestamp=None,
                evaluation_mode: EvaluationMode = EvaluationMode.EVALUATION,
                voxel_features: Optional[torch.Tensor] = None, **kwargs) -> Dict[str, Any]:
        image_features = kwargs.pop("image_features", None)
        rng_streams = kwargs.pop("rng_streams", None)
        training = evaluation_mode == EvaluationMode.TRAINING
        batch_size = len(camera)
        # number of target views (holo_diffusion_model.py:262-274): 1 in evaluation; in training the first
        # n_train_target_views cameras of the batch (all of them when <= 0), 1 when the batch is not larger than that
        if not training:
            n_targets = 1
        else:
            n_targets = batch_size if self.n_train_target_views <= 0 else min(self.n_train_target_views, batch_size)
        if batch_size <= n_targets:
            n_targets = 1
        target_cameras = camera[list(range(n_targets))]
        sampling_mode = RenderSamplingMode(self.sampling_mode_training if training else self.sampling_mode_evaluation)
        if not training and sampling_mode != RenderSamplingMode.FULL_GRID:
            raise NotImplementedError("evaluation uses full_grid sampling")
        if image_rgb is not None or image_features is not None:
            # ---- view pooling: views -> voxel grid (:327-374), one fused kernel behind the image feature extractor
            assert self.view_pooler_enabled, "view_pooler must be enabled to use image_rgb"
            assert voxel_features is None, "Cannot provide both image_rgb and voxel_features"
            # safe_slice_sources (:276-298): the source views are the frames of the FIRST frame's sequence minus the
            # n_targets leading ones; an empty selection falls back to the whole batch
            if sequence_name is not None and batch_size > 1:
                ok_ = [si for si, sname in enumerate(sequence_name) if sname == sequence_name[0]]
                sel = ok_[n_targets:]
            else:
                sel = list(range(n_targets, batch_size))
            if len(sel) == 0 or batch_size <= 1:
                sel = list(range(batch_size))
            if image_features is None:
                assert self.image_feature_extractor is not None, "Need an image_feature_extractor"
                image_features = self.image_feature_extractor(
                    image_rgb[sel], fg_probability[sel] if fg_probability is not None else None)
            else:
                # caller-supplied maps are the image feature extractor's output for the batch's source slots (frames
                # n_targets..) or for the whole batch; either way only the selected frames are pooled
                n_maps = int(next(iter(image_features.values())).shape[0])
                if n_maps == batch_size and batch_size > 1:
                    image_features = {k: v[sel] for k, v in image_features.items()}
                elif n_maps == batch_size - n_targets and sel != list(range(n_targets, batch_size)):
                    rel = [si - n_targets for si in sel]
                    image_features = {k: v[rel] for k, v in image_features.items()}
            source_cameras = camera[sel]
            voxel_features = self.pool_views_to_voxel_features(image_features, source_cameras)
        if voxel_features is None:
            assert not training, "training needs image_rgb / image_features or voxel_features"
            voxel_features = self.sample_random_voxel_features()
        if self.check_ranges:
            assert voxel_features.min() >= -1.0 and voxel_features.max() <= 1.0
        if self.net_3d_enabled:
            if self.diffusion_enabled and training:
                voxel_features = self._diffuse_and_denoise(voxel_features, rng_streams)
            else:
                voxel_features = self._refine(voxel_features)
        assert voxel_features.shape[1] == self.feature_size, "Wrong voxel feature size!"
        for func in self._implicit_functions:
            func.bind_args(voxel_grid_features=voxel_features)
        if training:
            rs = rng_streams or {}
            ray_bundle = self.raysampler(
                target_cameras, evaluation_mode, sampling_mode=sampling_mode, xys=rs.get("xys"),
                mask=mask_crop[list(range(n_targets))] if mask_crop is not None
                and sampling_mode == RenderSamplingMode.MASK_SAMPLE else None)
            rendered = self.renderer(ray_bundle=ray_bundle, implicit_functions=list(self._implicit_functions),
                                     evaluation_mode=evaluation_mode, rng_streams=rs)
        else:
            ray_bundle = self.raysampler(target_cameras, evaluation_mode, mask=None)
            rendered = self.renderer(ray_bundle=ray_bundle, implicit_functions=list(self._implicit_functions),
                                     evaluation_mode=evaluation_mode)
        for func in self._implicit_functions:
            func.unbind_args()
        preds: Dict[str, Any] = {"rendered": rendered, "ray_bundle": ray_bundle}
        preds["images_render"] = rendered.features.permute(0, 3, 1, 2)
        preds["depths_render"] = rendered.depths.permute(0, 3, 1, 2)
        preds["masks_render"] = rendered.masks.permute(0, 3, 1, 2)
        if rendered.normals is not None:
            # build-side addition: the reference renders the normals when the implicit function has render_normals=True
            # (released YAMLs) but never exports them; `normals_render` is the key its own fly-around output stage looks
            # for (flyaround.py:440-445, _make_shaded_from_normals)
            preds["normals_render"] = rendered.normals.permute(0, 3, 1, 2)
        return preds

    @torch.no_grad()
    def training_backward(self, *, camera: PerspectiveCameras, voxel_features: torch.Tensor, rng_streams: dict,
                          grads: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        """``loss.backward()`` of the TRAINING branch for a loss on the rendered outputs: the gradients that
        autograd leaves in the reference on the denoiser's and the RenderMLP's parameters and on the clean grid when
        ``forward(evaluation_mode=TRAINING)`` (holo_diffusion_model.py:384-457) is followed by losses on
        ``images_render / depths_render / masks_render`` and their prev_stage (coarse) counterparts (:458-489).
          rng_streams : the draws of the forward call - ``timesteps``, ``q_noise``, ``bootstrap`` (+ ``timesteps2``,
                        ``q_noise2``), ``xys`` and the renderer's ``u_coarse / u_fine / noise_coarse / noise_fine``
          grads       : d loss / d output, any of ``features (n_targets, n_rays, 1, 3)``, ``depths``, ``masks`` and
                        ``features_coarse`` / ``depths_coarse`` / ``masks_coarse``
        Chain: renderer backward (holo_render_rays_backward) -> the clamp of pred_xstart -> denoiser backward
        (holo_unet_backward) -> q_sample's sqrt(alpha_bar_t) -> (bootstrap) the first round's clamp and denoiser again; the
        parameter gradients of the two rounds add up, as the reference's shared ``net_3d`` accumulates them.
        Returns ``{"unet": {name: grad}, "render_mlp": {name: grad}, "voxel_features": grad of the clean grid,
        "voxel_grid": grad of the rendered grid}``.  When the clean grid came from the view-pooling branch (image_rgb inputs),
        ``pool_views_backward(image_features, source_cameras, out["voxel_features"])`` continues the chain to the mapper, the
        aggregator and the source-view feature maps."""
        assert self.net_3d_enabled and self.diffusion_enabled, "training_backward: the diffusion branch"
        rs = dict(rng_streams)
        for k in ("timesteps", "q_noise", "bootstrap", "xys"):
            if k not in rs:
                raise ValueError(f"training_backward: rng_streams['{k}'] is needed (the draw of the forward pass)")
        dev = voxel_features.device
        batch_size = len(camera)
        n_targets = batch_size if self.n_train_target_views <= 0 else min(self.n_train_target_views, batch_size)
        if batch_size <= n_targets:
            n_targets = 1
        target_cameras = camera[list(range(n_targets))]
        rounds = []

        def one_round(x0, t_key, n_key, last):
            t = torch.as_tensor(rs[t_key], device=dev, dtype=torch.int64).reshape(x0.shape[0])
            x_t = self.diffusion.q_sample(x0, t, noise=rs[n_key].to(dev))
            # the LAST round's forward is the taped one: its backward runs first and consumes the tape (no second forward);
            # an earlier round (bootstrap) is re-run by ``backward`` when its turn comes - the tape holds one forward
            y = self.net_3d.forward_train(x_t, t) if last else self.net_3d(x_t, t)
            rounds.append((x_t, t, y, last))
            return y.clamp(-1.0, 1.0)

        boot = bool(rs["bootstrap"])
        grid = one_round(voxel_features.float(), "timesteps", "q_noise", not boot)
        if boot:
            grid = one_round(grid, "timesteps2", "q_noise2", True)
        for func in self._implicit_functions:
            func.bind_args(voxel_grid_features=grid)
        try:
            bundle = self.raysampler(target_cameras, EvaluationMode.TRAINING,
                                     sampling_mode=RenderSamplingMode(self.sampling_mode_training), xys=rs["xys"])
            g_grid, render_grads = self.renderer.backward_training(bundle, list(self._implicit_functions), rs, grads)
        finally:
            for func in self._implicit_functions:
                func.unbind_args()
        unet_grads: Dict[str, torch.Tensor] = {}
        g = g_grid
        for x_t, t, y, taped in reversed(rounds):
            g_y = g * ((y >= -1.0) & (y <= 1.0)).to(g.dtype)  # torch.clamp passes the gradient inside [min, max]
            if taped:
                g_x, ug = self.net_3d.backward_taped(g_y)
            else:
                _, g_x, ug = self.net_3d.backward(x_t, t, g_y)
            for k, v in ug.items():
                unet_grads[k] = v if k not in unet_grads else unet_grads[k] + v
            g = self.diffusion._extract(self.diffusion.sqrt_alphas_cumprod, t, g_x.shape) * g_x  # d q_sample / d x_start
        return {"unet": unet_grads, "render_mlp": render_grads, "voxel_features": g, "voxel_grid": g_grid}

    def training_step(self, *, camera: PerspectiveCameras, voxel_features: torch.Tensor, rng_streams: dict, loss_fn) -> Dict[str, Any]:
        """One optimisation step's worth of gradients for an ARBITRARY torch loss on the rendered outputs - the role of
        ``preds["objective"].backward()`` in the reference's training loop (holo_diffusion_model.py:458-540, trainer/): the
        TRAINING forward runs on the HIP path, ``loss_fn(preds)`` is evaluated by torch on the small per-ray tensors
        (``images_render (n_targets,3,n_rays,1)``, ``depths_render``, ``masks_render`` and the ``*_coarse`` prev-stage
        counterparts, all leaves that require grad), torch's autograd yields d loss / d outputs, and
        ``training_backward`` carries them through the renderer and the denoiser.  Returns ``{"loss", "preds", "unet",
        "render_mlp", "voxel_features", "voxel_grid"}``.  ``rng_streams`` as for ``training_backward`` (all draws given).
        The reference's own objective (Implicitron ``ViewMetrics`` weighted by ``loss_weights``) is not restated here: any
        callable on these tensors takes its place."""
        with torch.no_grad():
            preds = self.forward(camera=camera, evaluation_mode=EvaluationMode.TRAINING, voxel_features=voxel_features,
                                 rng_streams=rng_streams)
        rend = preds["rendered"]
        leaves = {"images_render": rend.features.permute(0, 3, 1, 2), "depths_render": rend.depths.permute(0, 3, 1, 2),
                  "masks_render": rend.masks.permute(0, 3, 1, 2)}
        if rend.prev_stage is not None:
            leaves.update({"images_render_coarse": rend.prev_stage.features.permute(0, 3, 1, 2),
                           "depths_render_coarse": rend.prev_stage.depths.permute(0, 3, 1, 2),
                           "masks_render_coarse": rend.prev_stage.masks.permute(0, 3, 1, 2)})
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in leaves.items()}
        with torch.enable_grad():
            loss = loss_fn(dict(leaves))
            cots = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        names = {"images_render": "features", "depths_render": "depths", "masks_render": "masks",
                 "images_render_coarse": "features_coarse", "depths_render_coarse": "depths_coarse",
                 "masks_render_coarse": "masks_coarse"}
        grads = {names[k]: g.permute(0, 2, 3, 1).contiguous() for k, g in zip(leaves, cots) if g is not None}
        out = self.training_backward(camera=camera, voxel_features=voxel_features, rng_streams=rng_streams, grads=grads)
        out.update({"loss": loss.detach(), "preds": {k: v.detach() for k, v in leaves.items()}})
        return out

    def _diffuse_and_denoise(self, voxel_features: torch.Tensor, rng_streams: Optional[dict]) -> torch.Tensor:
        """The diffusion mechanism of the TRAINING branch (holo_diffusion_model.py:386-418): sample a timestep, diffuse the
        clean grid (q_sample), predict it back (pred_xstart of p_mean_variance, clamped) - and, with probability
        ``bootstrap_prob``, once more on the prediction ("bootstrap").  Random draws may be injected through
        ``rng_streams``: ``timesteps`` / ``q_noise`` (first round), ``bootstrap`` (bool), ``timesteps2`` / ``q_noise2``.
        With autograd enabled the rounds are differentiable (q_sample and the clamp in torch, the denoiser through its
        autograd node) - ``loss.backward()`` then reaches the denoiser's parameters and the clean grid as in the reference;
        otherwise the fused posterior kernel produces pred_xstart."""
        import numpy as np
        rs = rng_streams or {}
        dev = voxel_features.device

        def one_round(x0, t_key, n_key):
            t = rs.get(t_key)
            if t is None:
                t, _ = self.diffusion.sample_timesteps(x0.shape[0], dev)
            t = torch.as_tensor(t, device=dev, dtype=torch.int64).reshape(x0.shape[0])
            nz = rs.get(n_key)
            x_t = self.diffusion.q_sample(x0, t, noise=nz.to(dev) if nz is not None else None)
            if torch.is_grad_enabled() and (x_t.requires_grad or any(p.requires_grad for p in self.net_3d.parameters())):
                return self.net_3d(x_t, t).clamp(-1.0, 1.0)  # pred_xstart of the START_X parameterisation, clip_denoised
            return self.diffusion.p_mean_variance(model=self.net_3d, x=x_t, t=t, clip_denoised=True, model_kwargs={})["pred_xstart"]

        out = one_round(voxel_features, "timesteps", "q_noise")
        boot = rs.get("bootstrap")
        if boot is None:
            # the reference draws unconditionally and never reads enable_bootstrap here (holo_diffusion_model.py:402)
            boot = np.random.uniform() < self.bootstrap_prob
        if boot:
            out = one_round(out, "timesteps2", "q_noise2")
        return out

    @torch.no_grad()
    def pool_views_to_voxel_features(self, image_features: Dict[str, torch.Tensor], source_cameras) -> torch.Tensor:
        """tanh(pooled_feature_mapper(view_pooler(grid points, source views))) (:349-373) -> (1, F, R, R, R).
        ``image_features``: the image feature extractor's dict for the SOURCE views, key -> (n_src, C, H, W)."""
        assert self.view_pooler_enabled and self.view_pooler is not None, "view_pooler must be enabled"
        pm = self.pooled_feature_mapper
        A = self.view_pooler.get_aggregated_feature_dim(image_features)
        if isinstance(pm.weight, torch.nn.parameter.UninitializedParameter):  # first use of the LazyLinear (Xavier, :37-41)
            dev = next(iter(image_features.values())).device
            pm.in_features = A
            pm.weight.materialize((self.feature_size, A), device=dev)
            pm.bias.materialize((self.feature_size,), device=dev)
            torch.nn.init.xavier_uniform_(pm.weight.data)
            pm.bias.data.zero_()
            pm.__class__ = torch.nn.Linear
        return self.view_pooler.pool_to_voxel_features(image_features, source_cameras, pm.weight, pm.bias, self.resol,
                                                       self.volume_extent)

    def pool_views_backward(self, image_features: Dict[str, torch.Tensor], source_cameras,
                            grad_voxel_features: torch.Tensor, want_feature_grads: bool = True) -> Dict[str, Any]:
        """The encoder side of ``loss.backward()`` (holo_diffusion_model.py:340-373 under autograd): the gradient of the
        pooled grid - ``training_backward(...)["voxel_features"]`` when the grid came from ``pool_views_to_voxel_features`` -
        through tanh, ``pooled_feature_mapper`` and the view pooling.  Returns ``{"image_features": {key: grad},
        "pooled_feature_mapper": {"weight": grad, "bias": grad}}`` (+ ``"feature_aggregator": {name: grad}`` for the learnt
        MLPMeanFeatureAggregator); the feature-map gradients are what autograd hands the image feature extractor (which is
        outside this path)."""
        assert self.view_pooler_enabled and self.view_pooler is not None, "view_pooler must be enabled"
        pm = self.pooled_feature_mapper
        if isinstance(pm.weight, torch.nn.parameter.UninitializedParameter):
            raise RuntimeError("pool_views_backward: pooled_feature_mapper is not materialised (run the forward first)")
        gf, gw, gb = self.view_pooler.pool_to_voxel_features_backward(
            image_features, source_cameras, pm.weight, pm.bias, self.resol, self.volume_extent, grad_voxel_features,
            want_feature_grads=want_feature_grads)
        out = {"image_features": gf, "pooled_feature_mapper": {"weight": gw, "bias": gb}}
        agg_grads = getattr(self.view_pooler.feature_aggregator, "native_grads", None)
        if agg_grads is not None:  # MLPMeanFeatureAggregator: its own parameters, by reference name
            out["feature_aggregator"] = dict(agg_grads)
        return out

    def render_views(self, voxel_features: torch.Tensor, cameras: PerspectiveCameras) -> Dict[str, torch.Tensor]:
        """Batched turntable render: all cameras of a fly-around in ONE holo_render call (BASELINE config 4).
        Equivalent to calling forward() once per camera with the same voxel_features."""
        if self.net_3d_enabled:
            voxel_features = self._refine(voxel_features)
        for func in self._implicit_functions:
            func.bind_args(voxel_grid_features=voxel_features)
        bundle = self.raysampler(cameras, EvaluationMode.EVALUATION)
        rendered = self.renderer(ray_bundle=bundle, implicit_functions=list(self._implicit_functions),
                                 evaluation_mode=EvaluationMode.EVALUATION)
        for func in self._implicit_functions:
            func.unbind_args()
        out = {"images_render": rendered.features.permute(0, 3, 1, 2),
               "depths_render": rendered.depths.permute(0, 3, 1, 2),
               "masks_render": rendered.masks.permute(0, 3, 1, 2)}
        if rendered.normals is not None:
            out["normals_render"] = rendered.normals.permute(0, 3, 1, 2)
        return out
