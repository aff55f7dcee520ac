"""CPU restatement (torch CPU ops, fp32) of the view-pooling entry of the reference: source-view feature maps ->
voxel feature grid.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  PARITY UNPINNED: the arithmetic below the reference's call site
lives in PyTorch3D 0.7.4 (un-vendored, not installable here); it is restated from the published source.

  reference call site      holo_diffusion/holo_diffusion_model.py:327-374
      grid_xyz = VolumeLocator(...).get_coord_grid()                     (:349-356)
      voxel_features = self.view_pooler(pts=grid_xyz, camera=..., feats=img_feats, masks=mask_crop)   (:358-367)
      voxel_features = tanh(pooled_feature_mapper(voxel_features))       (:368-373;  LazyLinear -> feature_size, :113)
  released configuration   configs/apple.yaml:183-196 (same in every released YAML):
      view_sampler: masked_sampling false, sampling_mode bilinear
      feature_aggregator: AngleWeightedReductionFeatureAggregator, reduction_functions [AVG, STD],
                          weight_by_ray_angle_gamma 1.0, min_ray_angle_weight 0.1;
                          exclude_target_view / exclude_target_view_mask_features forced to False (:114-116)

  PyTorch3D algorithms restated:
    * ``VolumeLocator.get_coord_grid``: voxel centres, local coordinates linspace(-1, 1, R) per axis scaled by the
      half extent 0.5 (R-1) voxel_size (the same locator the renderer's trilinear fetch uses), x fastest
    * ``ViewSampler`` / ``project_points_and_sample``: ``camera.transform_points`` (PerspectiveCameras in NDC:
      X_cam = X R + T, ndc = f * X_cam.xy / z + p, |z| clamped to eps = 1e-2 keeping the sign... NOTE: the reference
      passes PyTorch3D's default eps), ``ndc_grid_sample`` = ``F.grid_sample(feat, -ndc (long side divided by the
      aspect ratio), mode=bilinear, padding zeros, align_corners=False)``; masks all ones (masked_sampling false)
    * ``_get_point_to_source_camera_ray_dirs`` is the REFERENCE's own copy (custom_modules.py:279-334): normalize(pts - C)
    * ``_get_angular_reduction_weights``: w = clamp((0.5 (d_v . d_0 + 1)) ** gamma, min_ray_angle_weight) * mask
    * ``_avgmaxstd_reduction_function``: AVG = wmean(x, w, eps 1e-2); STD = sqrt(clamp(wmean((x - AVG)^2, w, eps 1e-2), 1e-4));
      ``wmean(x, w) = sum(x w) / clamp(sum w, eps)``; outputs concatenated per feature key as [AVG | STD], keys in dict order
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def coord_grid(resol: int, volume_extent: float) -> torch.Tensor:
    """World coordinates of the voxel centres, (R^3, 3), voxel v = (z*R + y)*R + x."""
    half = 0.5 * (resol - 1) * (volume_extent / resol)
    lin = torch.linspace(-1.0, 1.0, resol, dtype=torch.float32) * half
    Z, Y, X = torch.meshgrid(lin, lin, lin, indexing="ij")
    return torch.stack([X, Y, Z], dim=-1).reshape(-1, 3)


def project_ndc(pts: torch.Tensor, R: torch.Tensor, T: torch.Tensor, focal: torch.Tensor, pp: torch.Tensor,
                eps: float = 1e-2) -> torch.Tensor:
    """PerspectiveCameras.transform_points (NDC): pts (P,3), one camera -> (P,2)."""
    cam = pts @ R + T[None]
    z = cam[:, 2]
    zc = torch.where(z.abs() < eps, torch.where(z < 0, -torch.ones_like(z), torch.ones_like(z)) * eps, z)
    return torch.stack([focal[0] * cam[:, 0] / zc + pp[0], focal[1] * cam[:, 1] / zc + pp[1]], dim=-1)


def ndc_grid_sample(feat: torch.Tensor, ndc: torch.Tensor) -> torch.Tensor:
    """feat (C,H,W), ndc (P,2) -> (P,C): bilinear, zeros padding, align_corners=False."""
    C, H, W = feat.shape
    g = -ndc.clone()
    if W >= H:
        g[:, 0] = g[:, 0] / (W / H)
    else:
        g[:, 1] = g[:, 1] / (H / W)
    out = F.grid_sample(feat[None], g[None, :, None, :], mode="bilinear", padding_mode="zeros", align_corners=False)
    return out[0, :, :, 0].t()


def ray_dirs_to_cameras(pts: torch.Tensor, R: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """custom_modules.py:279-334: normalize(pts - camera centre), (P,3) for one camera."""
    centre = -(T[None] @ R.t())[0]
    return F.normalize(pts - centre[None], dim=-1)


def pool_views(feats: Dict[str, torch.Tensor], cams: dict, resol: int, volume_extent: float, gamma: float = 1.0,
               min_weight: float = 0.1, eps_proj: float = 1e-2) -> torch.Tensor:
    """feats: key -> (n_src, C_k, H_k, W_k); cams: R (n,3,3), T (n,3), focal (n,2), pp (n,2).
    Returns the aggregated features (R^3, 2 * sum C_k): per key [AVG | STD], keys in dict order."""
    pts = coord_grid(resol, volume_extent)
    n = cams["R"].shape[0]
    dirs = torch.stack([ray_dirs_to_cameras(pts, cams["R"][v], cams["T"][v]) for v in range(n)])      # (n,P,3)
    dots = (dirs * dirs[:1]).sum(-1)                                                                    # vs view 0
    w = (0.5 * (dots + 1.0)).pow(gamma).clamp(min_weight)                                               # (n,P); masks = 1
    ndc = torch.stack([project_ndc(pts, cams["R"][v], cams["T"][v], cams["focal"][v], cams["pp"][v], eps_proj)
                       for v in range(n)])                                                              # (n,P,2)
    outs = []
    for k, f in feats.items():
        x = torch.stack([ndc_grid_sample(f[v], ndc[v]) for v in range(n)])                             # (n,P,C)
        den = w.sum(0).clamp(1e-2)[:, None]
        mu = (x * w[..., None]).sum(0) / den
        var = (((x - mu[None]) ** 2) * w[..., None]).sum(0) / den
        outs += [mu, var.clamp(1e-4).sqrt()]
    return torch.cat(outs, dim=-1)


def voxel_features_from_views(feats: Dict[str, torch.Tensor], cams: dict, mapper_w: torch.Tensor,
                              mapper_b: torch.Tensor, resol: int, volume_extent: float, **kw) -> torch.Tensor:
    """holo_diffusion_model.py:358-373: tanh(pooled_feature_mapper(view_pooler(...))) -> (1, F, R, R, R)."""
    agg = pool_views(feats, cams, resol, volume_extent, **kw)
    vf = F.linear(agg, mapper_w, mapper_b)                 # (R^3, F)
    return torch.tanh(vf.t().reshape(1, -1, resol, resol, resol))


# ---------------------------------------------------------------------------------------------------------------------
# MLPMeanFeatureAggregator (the REFERENCE's own aggregator, custom_modules.py:162-293; selected by configs/hydrant.yaml:184
# and old_base_config.yaml:205).  Its body is plain torch and is executed from the reference source by
# oracle/make_golden_render.py (fixtures tests/golden/ref_mlp_mean_aggregator.npz); the sampling in front of it
# (ViewSampler) is PyTorch3D's and stays restated.
# ---------------------------------------------------------------------------------------------------------------------
def harmonic_embedding(x: torch.Tensor, n: int) -> torch.Tensor:
    freqs = 2.0 ** torch.arange(n, dtype=torch.float32)
    e = (x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    return torch.cat((e.sin(), e.cos(), x), dim=-1)


def mlp_mean_param_shapes(in_dim: int, n_hidden: int = 128, dim_out: int = 128, prefix: str = "") -> Dict[str, tuple]:
    """State-dict names of MLPMeanFeatureAggregator (custom_modules.py:179-196); ``in_dim`` = sum of the sampled
    feature channels + 3 (2 n_harmonic_functions_ray + 1) (the two LazyLinear layers take their width from it)."""
    p = prefix
    return {p + "_first_sampled.weight": (n_hidden, in_dim), p + "_first_sampled.bias": (n_hidden,),
            p + "_first_mean.weight": (n_hidden, in_dim), p + "_first_mean.bias": (n_hidden,),
            p + "_last.weight": (dim_out, n_hidden), p + "_last.bias": (dim_out,),
            p + "_mlp.mlp.0.0.weight": (n_hidden, n_hidden), p + "_mlp.mlp.0.0.bias": (n_hidden,)}


def mlp_mean_aggregate(feats_sampled: Sequence[torch.Tensor], ray_dirs: torch.Tensor, weights: torch.Tensor,
                       sd: Dict[str, torch.Tensor], n_harmonic: int = 3, prefix: str = "") -> torch.Tensor:
    """``_mlp_pass`` (custom_modules.py:243-262) for one voxel batch.  feats_sampled: list of (n_src, P, C_k); ray_dirs
    (n_src, P, 3) unit vectors point -> ... from the camera centres; weights (n_src, P).  Returns (P, dim_out)."""
    x = torch.cat(list(feats_sampled) + [harmonic_embedding(ray_dirs, n_harmonic)], dim=-1) * weights[..., None]
    mean = (x * weights[..., None]).sum(0, keepdim=True) / weights[..., None].sum(0, keepdim=True).clamp(1e-2)  # wmean
    p = prefix
    mlp_in = F.linear(x, sd[p + "_first_sampled.weight"], sd[p + "_first_sampled.bias"]) \
        + F.linear(mean, sd[p + "_first_mean.weight"], sd[p + "_first_mean.bias"])
    # MLPWithInputSkips(n_layers=1): the single layer IS the last one -> Linear + LeakyReLU(0.2) (custom_modules.py:108-112)
    h = F.leaky_relu(F.linear(mlp_in, sd[p + "_mlp.mlp.0.0.weight"], sd[p + "_mlp.mlp.0.0.bias"]), 0.2)
    out = F.linear(h, sd[p + "_last.weight"], sd[p + "_last.bias"])
    return (out * torch.softmax(out[..., :1], dim=0)).sum(dim=0)


def pool_views_mlp_mean(feats: Dict[str, torch.Tensor], cams: dict, sd: Dict[str, torch.Tensor], resol: int,
                        volume_extent: float, n_harmonic: int = 3, eps_proj: float = 1e-2, prefix: str = "") -> torch.Tensor:
    """ViewSampler (restated) + MLPMeanFeatureAggregator at the voxel centres: (R^3, dim_out)."""
    pts = coord_grid(resol, volume_extent)
    n = cams["R"].shape[0]
    dirs = torch.stack([ray_dirs_to_cameras(pts, cams["R"][v], cams["T"][v]) for v in range(n)])
    ndc = torch.stack([project_ndc(pts, cams["R"][v], cams["T"][v], cams["focal"][v], cams["pp"][v], eps_proj)
                       for v in range(n)])
    sampled = [torch.stack([ndc_grid_sample(f[v], ndc[v]) for v in range(n)]) for f in feats.values()]
    w = torch.ones(n, pts.shape[0])  # masked_sampling false -> masks 1; exclude_target_view forced False (:114-116)
    return mlp_mean_aggregate(sampled, dirs, w, sd, n_harmonic, prefix)


def voxel_features_from_views_mlp_mean(feats, cams, sd, mapper_w, mapper_b, resol, volume_extent, **kw) -> torch.Tensor:
    agg = pool_views_mlp_mean(feats, cams, sd, resol, volume_extent, **kw)
    return torch.tanh(F.linear(agg, mapper_w, mapper_b).t().reshape(1, -1, resol, resol, resol))
