"""CPU restatement (torch CPU ops, fp32) of the HoloDiffusion render path.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  PARITY UNPINNED: the reference
files on this half cannot be imported (they need PyTorch3D 0.7.4, un-vendored,
``/root/reference/environment.yaml:139``).  This file restates

  reference call sites (relative to /root/reference/holo_diffusion):
    * ``holo_voxel_grid_implicit_function.py:107-129``  RenderMLP.forward
    * ``holo_voxel_grid_implicit_function.py:182-269``  HoloVoxelGridImplicitFunction.forward
    * ``custom_modules.py:61-160``  MLPWithInputSkips (incl. the :108-112 quirk: the hidden
      activation is attached to the LAST layer only, every other layer gets ``last_activation``
      = identity)
    * ``holo_multipass_ea.py:79-125``  coarse pass -> refiner -> fine pass
    * ``holo_diffusion_model.py:442-457,515-523``  ray sampler call, _render, output permutes
    * ``utils/render_utils/flyaround.py:301-350``  get_simple_360_camera_trajectory
    * ``configs/apple.yaml:135-165``  sampler / refiner / raymarcher settings

  PyTorch3D 0.7.4 algorithms (restated from the published source, not present in-tree):
    * ``NDCMultinomialRaysampler`` / ``_xy_to_ray_bundle``: NDC pixel-centre grid, unprojection
      of the z=1 and z=2 planes, ``directions = p2 - p1``, ``origins = p1 - directions``
    * ``AdaptiveRaySampler`` depth bounds (``get_min_max_depth_bounds``)
    * ``PerspectiveCameras`` (NDC, row-vector convention ``X_cam = X_world R + T``)
    * ``VolumeLocator.world_to_local_coords`` + ``FullResolutionVoxelGrid`` -> ``F.grid_sample``
      (bilinear, zeros padding, align_corners=True)
    * ``HarmonicEmbedding`` (logspace, append_input)
    * ``EmissionAbsorptionRaymarcher`` (surface_thickness 1, background_opacity 1e10,
      density_relu, blend_output False, bg (1,1,1))
    * ``RayPointRefiner`` + ``sample_pdf`` (deterministic u = linspace(0,1,n), eps 1e-5).
      FIDELITY CAVEAT: PyTorch3D's production refiner calls the compiled ``_C.sample_pdf`` - a sequential fp32 running
      sum of the normalised weights per ray and a ``bin_weight > eps`` branch for the in-bin interpolation - while this
      file restates the library's documented equivalent ``sample_pdf_python`` (``torch.cumsum``, ``denom < eps -> 1``).
      The two agree except on which side of the eps switch an (almost) empty bin falls, i.e. exactly on the "fragile
      rays" that ``tests/test_gpu_configs.py`` isolates (a raw ``denom`` within 0.3 eps of the switch); the HIP kernel
      follows ``sample_pdf_python`` too (double running sums rounded to fp32, like torch's CPU cumsum).
    * ``look_at_view_transform`` / ``so3_exp_map``
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class RenderCfg:
    resol: int = 64
    feature_size: int = 32
    volume_extent: float = 8.0
    scene_extent: float = 4.0
    scene_center: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    n_pts_coarse: int = 64
    n_pts_fine: int = 64
    image_height: int = 400
    image_width: int = 400
    bg_color: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    background_opacity: float = 1e10
    dnet_hidden_dim: int = 256
    dir_emb_dims: int = 4
    sample_pdf_eps: float = 1e-5
    feature_dim: int = 0  # RenderMLP.output_vp_independent_feature_dims (0 inside HoloDiffusionModel, 64 by default)


# ----------------------------------------------------------------------------
# RenderMLP parameters
# ----------------------------------------------------------------------------
def render_mlp_param_shapes(cfg: RenderCfg, prefix: str = "") -> Dict[str, Tuple[int, ...]]:
    """Reference names below ``..._fn.render_mlp.`` (holo_voxel_grid_implicit_function.py:73-92)."""
    C, Hd = cfg.feature_size, cfg.dnet_hidden_dim
    demb = 3 * (2 * cfg.dir_emb_dims + 1)
    p = prefix
    shapes = {
        p + "_density_net.mlp.0.0.weight": (Hd, C), p + "_density_net.mlp.0.0.bias": (Hd,),
        p + "_density_net.mlp.1.0.weight": (Hd, Hd), p + "_density_net.mlp.1.0.bias": (Hd,),
        p + "_density_net.mlp.2.0.weight": (Hd, Hd + C), p + "_density_net.mlp.2.0.bias": (Hd,),
        p + "_density_net.mlp.3.0.weight": (Hd + 1, Hd), p + "_density_net.mlp.3.0.bias": (Hd + 1,),
        p + "_radiance_net.mlp.0.0.weight": (3, Hd + demb), p + "_radiance_net.mlp.0.0.bias": (3,),
    }
    if cfg.feature_dim > 0:  # the view-point independent feature head (holo_voxel_grid_implicit_function.py:94-105)
        shapes[p + "_feature_net.mlp.0.0.weight"] = (cfg.feature_dim, Hd)
        shapes[p + "_feature_net.mlp.0.0.bias"] = (cfg.feature_dim,)
    return shapes


# ----------------------------------------------------------------------------
# cameras (PyTorch3D conventions)
# ----------------------------------------------------------------------------
def look_at_view_transform(dist: float, elev_deg: float, azim_deg: float,
                           up=(0.0, 1.0, 0.0)) -> Tuple[torch.Tensor, torch.Tensor]:
    elev = math.pi / 180.0 * torch.tensor([float(elev_deg)])
    azim = math.pi / 180.0 * torch.tensor([float(azim_deg)])
    d = torch.tensor([float(dist)])
    C = torch.stack([d * torch.cos(elev) * torch.sin(azim), d * torch.sin(elev),
                     d * torch.cos(elev) * torch.cos(azim)], dim=1)          # (1,3)
    at = torch.zeros(1, 3)
    upv = torch.tensor([up], dtype=torch.float32)
    z_axis = F.normalize(at - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(upv, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        repl = F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5)
        x_axis = torch.where(is_close, repl, x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T


def so3_exp_map(log_rot: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    nrms = (log_rot * log_rot).sum(1)
    ang = torch.clamp(nrms, eps).sqrt()
    inv = 1.0 / ang
    fac1 = inv * ang.sin()
    fac2 = inv * inv * (1.0 - ang.cos())
    K = torch.zeros(log_rot.shape[0], 3, 3)
    x, y, z = log_rot.unbind(1)
    K[:, 0, 1], K[:, 0, 2] = -z, y
    K[:, 1, 0], K[:, 1, 2] = z, -x
    K[:, 2, 0], K[:, 2, 1] = -y, x
    K2 = torch.bmm(K, K)
    return fac1[:, None, None] * K + fac2[:, None, None] * K2 + torch.eye(3)[None]


def simple_360_cameras(n_poses: int, elevation_rad: float = -30.0 * (2 * math.pi / 360), radius: float = 10.0,
                       up=(0.0, -1.0, 0.0), focal: float = 3.2, max_angle: float = 2 * math.pi,
                       canonical_up=(0.0, -1.0, 0.0)):
    """flyaround.py:301-350.  Returns dict R (n,3,3), T (n,3), focal (n,2), pp (n,2)."""
    max_angle_deg = 360 * max_angle / (math.pi * 2)
    elev_deg = 360 * elevation_rad / (math.pi * 2)
    azimuths = torch.linspace(0, max_angle_deg, n_poses + 1)[:n_poses]
    Rs, Ts = [], []
    for az in azimuths:
        R, T = look_at_view_transform(radius, elev_deg, float(az), up=canonical_up)
        Rs.append(R)
        Ts.append(T)
    Rs, Ts = torch.cat(Rs), torch.cat(Ts)
    axis = torch.cross(torch.tensor(canonical_up, dtype=torch.float32), torch.tensor(up, dtype=torch.float32), dim=0)
    R_plane = so3_exp_map(axis[None])[0]
    Rs = torch.bmm(R_plane[None].expand_as(Rs), Rs)
    return {"R": Rs, "T": Ts, "focal": torch.full((n_poses, 2), focal), "pp": torch.zeros(n_poses, 2)}


# ----------------------------------------------------------------------------
# rays
# ----------------------------------------------------------------------------
def ndc_pixel_grid(H: int, W: int) -> torch.Tensor:
    """NDCMultinomialRaysampler xy grid, (H, W, 2) with [...,0]=x, [...,1]=y."""
    if W >= H:
        range_x, range_y = W / H, 1.0
    else:
        range_x, range_y = 1.0, H / W
    hx, hy = range_x / W, range_y / H
    xs = torch.linspace(range_x - hx, -range_x + hx, W, dtype=torch.float32)
    ys = torch.linspace(range_y - hy, -range_y + hy, H, dtype=torch.float32)
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([X, Y], dim=-1)


def depth_bounds(R: torch.Tensor, T: torch.Tensor, cfg: RenderCfg) -> Tuple[float, float]:
    """AdaptiveRaySampler: bounds from the camera centre C = -T R^T."""
    C = -(T[None] @ R.t())[0]
    sc = torch.tensor(cfg.scene_center, dtype=torch.float32)
    dist = ((C - sc) ** 2).sum().clamp(0.001).sqrt().clamp(cfg.scene_extent + 1e-3)
    return float(dist - cfg.scene_extent), float(dist + cfg.scene_extent)


def make_rays(cam: dict, cfg: RenderCfg):
    """Returns origins (H*W,3), directions (H*W,3), lengths (H*W,P) for one camera (index 0 of ``cam``)."""
    R, T = cam["R"].reshape(-1, 3, 3)[0].float(), cam["T"].reshape(-1, 3)[0].float()
    f, pp = cam["focal"].reshape(-1, 2)[0].float(), cam["pp"].reshape(-1, 2)[0].float()
    H, W = cfg.image_height, cfg.image_width
    xy = ndc_pixel_grid(H, W).reshape(-1, 2)
    d_cam = torch.stack([(xy[:, 0] - pp[0]) / f[0], (xy[:, 1] - pp[1]) / f[1], torch.ones(xy.shape[0])], dim=-1)
    p1 = (d_cam * 1.0 - T[None]) @ R.t()
    p2 = (d_cam * 2.0 - T[None]) @ R.t()
    dirs = p2 - p1
    origins = p1 - dirs
    zmin, zmax = depth_bounds(R, T, cfg)
    lengths = torch.linspace(zmin, zmax, cfg.n_pts_coarse, dtype=torch.float32)[None].expand(xy.shape[0], -1)
    return origins, dirs, lengths


# ----------------------------------------------------------------------------
# implicit function
# ----------------------------------------------------------------------------
def trilinear(grid: torch.Tensor, pts_world: torch.Tensor, cfg: RenderCfg) -> torch.Tensor:
    """grid (1,C,D,H,W), pts (...,3) world -> (...,C).  VolumeLocator + grid_sample."""
    voxel_size = cfg.volume_extent / cfg.resol
    half = 0.5 * (cfg.resol - 1) * voxel_size
    local = pts_world / half
    shp = local.shape[:-1]
    out = F.grid_sample(grid, local.reshape(1, -1, 1, 1, 3), mode="bilinear", padding_mode="zeros",
                        align_corners=True)
    return out.reshape(grid.shape[1], -1).t().reshape(*shp, grid.shape[1])


def harmonic_embedding(x: torch.Tensor, n: int) -> torch.Tensor:
    freqs = 2.0 ** torch.arange(n, dtype=torch.float32)
    e = (x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    return torch.cat((e.sin(), e.cos(), x), dim=-1)


def render_mlp(sd: Dict[str, torch.Tensor], feats: torch.Tensor, dirs_normed: torch.Tensor, cfg: RenderCfg,
               prefix: str = ""):
    """RenderMLP.forward in the reference (uncollapsed) formulation.  feats (...,C), dirs (...,3)."""
    p = prefix + "_density_net.mlp."
    y = F.linear(feats, sd[p + "0.0.weight"], sd[p + "0.0.bias"])
    y = F.linear(y, sd[p + "1.0.weight"], sd[p + "1.0.bias"])
    y = torch.cat((y, feats), dim=-1)
    y = F.linear(y, sd[p + "2.0.weight"], sd[p + "2.0.bias"])
    y = F.linear(y, sd[p + "3.0.weight"], sd[p + "3.0.bias"])
    y = F.leaky_relu(y, 0.2)
    mlp_feats, dens = y[..., :-1], y[..., -1:]
    r = prefix + "_radiance_net.mlp.0.0."
    e = harmonic_embedding(dirs_normed, cfg.dir_emb_dims)
    rad = F.leaky_relu(F.linear(torch.cat([mlp_feats, e], dim=-1), sd[r + "weight"], sd[r + "bias"]), 0.2)
    return dens, torch.sigmoid(rad)


def render_mlp_vp_features(sd: Dict[str, torch.Tensor], feats: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """The third output of RenderMLP.forward (holo_voxel_grid_implicit_function.py:125-129): ``_feature_net`` - one Linear
    (rnet_num_layers = 1) followed by the LeakyReLU the construction quirk attaches to a LAST layer - on the density
    net's hidden features."""
    p = prefix + "_density_net.mlp."
    y = F.linear(feats, sd[p + "0.0.weight"], sd[p + "0.0.bias"])
    y = F.linear(y, sd[p + "1.0.weight"], sd[p + "1.0.bias"])
    y = torch.cat((y, feats), dim=-1)
    y = F.linear(y, sd[p + "2.0.weight"], sd[p + "2.0.bias"])
    y = F.leaky_relu(F.linear(y, sd[p + "3.0.weight"], sd[p + "3.0.bias"]), 0.2)
    f = prefix + "_feature_net.mlp.0.0."
    return F.leaky_relu(F.linear(y[..., :-1], sd[f + "weight"], sd[f + "bias"]), 0.2)


def implicit_function_pts(grid, sd, pts, cfg: RenderCfg, prefix: str = ""):
    """The ``pts_3d`` entry of HoloVoxelGridImplicitFunction.forward (holo_voxel_grid_implicit_function.py:182-269):
    arbitrary world points (...,3), dummy all-ones directions (normalised) -> densities (...,1), features
    (..., 3 + feature_dim) = [colour | view-point independent features]."""
    feats = trilinear(grid, pts, cfg)
    dn = F.normalize(torch.ones_like(pts), dim=-1)
    dens, col = render_mlp(sd, feats, dn, cfg, prefix)
    if cfg.feature_dim > 0:
        col = torch.cat([col, render_mlp_vp_features(sd, feats, prefix)], dim=-1)
    return dens, col


def implicit_function(grid, sd, origins, dirs, lengths, cfg: RenderCfg, prefix: str = ""):
    """(rays,3),(rays,3),(rays,P) -> densities (rays,P,1), colours (rays,P,3)."""
    pts = origins[:, None, :] + lengths[:, :, None] * dirs[:, None, :]
    feats = trilinear(grid, pts, cfg)
    dn = F.normalize(dirs, dim=-1)[:, None, :].expand(-1, lengths.shape[1], -1)
    return render_mlp(sd, feats, dn, cfg, prefix)


def implicit_normals(grid, sd, pts, cfg: RenderCfg, prefix: str = ""):
    """RenderMLP.get_normals (holo_voxel_grid_implicit_function.py:131-145) at world points pts (...,3): autograd of
    the summed density (last output of the density net, after its LeakyReLU) w.r.t. the points, F.normalize'd."""
    with torch.enable_grad():
        x = pts.clone().requires_grad_(True)
        dens, _ = render_mlp(sd, trilinear(grid, x, cfg), torch.zeros_like(x), cfg, prefix)
        (g,) = torch.autograd.grad(dens.sum(), x)
    return F.normalize(g, dim=-1)


# ----------------------------------------------------------------------------
# raymarcher + refiner
# ----------------------------------------------------------------------------
def ea_raymarch(dens: torch.Tensor, feats: torch.Tensor, lengths: torch.Tensor, cfg: RenderCfg,
                noise: Optional[torch.Tensor] = None, noise_std: float = 0.0):
    """``noise`` (rays, P) standard normals: the raymarcher's ``density_noise_std`` branch (training mode,
    holo_multipass_ea.py:87-91): densities + randn_like * std BEFORE the ReLU, with the draw injected."""
    deltas = torch.cat((torch.diff(lengths, dim=-1),
                        torch.full_like(lengths[..., :1], cfg.background_opacity)), dim=-1)
    raw = dens[..., 0]
    if noise is not None and noise_std > 0.0:
        raw = raw + noise * noise_std
    d = torch.relu(raw)
    wd = deltas * d
    capped = 1.0 - torch.exp(-wd)
    ray_op = 1.0 - torch.exp(-torch.cumsum(wd, dim=-1))
    opac = ray_op[..., -1:]
    absorb = (-ray_op + 1.0).roll(1, dims=-1)
    absorb[..., :1] = 1.0
    w = capped * absorb
    rgb = (w[..., None] * feats).sum(dim=-2)
    depth = (w * lengths)[..., None].sum(dim=-2)
    bg = torch.tensor(cfg.bg_color, dtype=torch.float32)
    rgb = rgb + (1 - opac) * bg
    return rgb, depth, opac, w


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int, eps: float = 1e-5,
               diag: Optional[dict] = None, u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Deterministic inverse-CDF sampling (pytorch3d sample_pdf, det=True).  ``diag`` (test diagnostics) receives the
    raw ``denom = cdf_above - cdf_below`` of every sample BEFORE the ``denom < eps -> 1`` switch."""
    weights = weights + eps
    pdf = weights / weights.sum(dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:  # det = True
        u = torch.linspace(0.0, 1.0, n_samples, dtype=cdf.dtype).expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()
    else:          # det = False (training): u = torch.rand(...), injected
        u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    if diag is not None:
        diag["denom"] = denom.clone()
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def jiggle_within_stratas(bin_centers: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """PyTorch3D ``_jiggle_within_stratas`` with the uniform draw injected: one sample per stratum, strata bounded by
    the mid points of neighbouring depths (and the first / last depth itself)."""
    mids = 0.5 * (bin_centers[..., 1:] + bin_centers[..., :-1])
    upper = torch.cat((mids, bin_centers[..., -1:]), dim=-1)
    lower = torch.cat((bin_centers[..., :1], mids), dim=-1)
    return lower + (upper - lower) * u


def rays_from_xys(cam: dict, xys: torch.Tensor, cfg: RenderCfg):
    """Rays of ONE camera through the NDC points xys (n,2) (mask-sampled rays of the training branch): the same
    un-projection as the full grid (make_rays)."""
    R, T = cam["R"].reshape(-1, 3, 3)[0].float(), cam["T"].reshape(-1, 3)[0].float()
    f, pp = cam["focal"].reshape(-1, 2)[0].float(), cam["pp"].reshape(-1, 2)[0].float()
    d_cam = torch.stack([(xys[:, 0] - pp[0]) / f[0], (xys[:, 1] - pp[1]) / f[1], torch.ones(xys.shape[0])], dim=-1)
    p1 = (d_cam * 1.0 - T[None]) @ R.t()
    p2 = (d_cam * 2.0 - T[None]) @ R.t()
    dirs = p2 - p1
    zmin, zmax = depth_bounds(R, T, cfg)
    lengths = torch.linspace(zmin, zmax, cfg.n_pts_coarse, dtype=torch.float32)[None].expand(xys.shape[0], -1)
    return p1 - dirs, dirs, lengths


def refine_lengths(lengths: torch.Tensor, weights: torch.Tensor, cfg: RenderCfg, diag: Optional[dict] = None,
                   u: Optional[torch.Tensor] = None) -> torch.Tensor:
    mid = torch.lerp(lengths[..., 1:], lengths[..., :-1], 0.5)
    z = sample_pdf(mid, weights[..., 1:-1], cfg.n_pts_fine, cfg.sample_pdf_eps, diag, u)
    return torch.sort(torch.cat((lengths, z), dim=-1), dim=-1)[0]


@torch.no_grad()
def render_rays(grid: torch.Tensor, sd: Dict[str, torch.Tensor], origins: torch.Tensor, dirs: torch.Tensor,
                lengths: torch.Tensor, cfg: RenderCfg, prefix: str = "", chunk_rays: int = 4096,
                with_normals: bool = False, u_coarse: Optional[torch.Tensor] = None, u_fine: Optional[torch.Tensor] = None,
                noise_coarse: Optional[torch.Tensor] = None, noise_fine: Optional[torch.Tensor] = None,
                noise_std: float = 0.0, fine_lengths: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """``fine_lengths`` (n, P + Pf): use this merged depth list for the fine pass instead of the refiner's (tests that
    hold the sample placement of the implementation under test fixed).
    Two-pass render of an ARBITRARY set of rays (origins (n,3), directions (n,3), coarse lengths (n,P)): coarse pass
    -> refiner -> fine pass (holo_multipass_ea.py:79-125).  Per-ray outputs: rgb (n,3), depth (n,1), mask (n,1), the
    coarse-pass rgb_c / depth_c / mask_c and the sorted fine lengths; with ``with_normals`` also the rendered normals
    of both passes, sum_i w_i n_i (holo_multipass_ea.py:105-109)."""
    keys = ["rgb", "depth", "mask", "rgb_c", "depth_c", "mask_c", "fine_lengths", "pdf_denom"] + (["normals", "normals_c"] if with_normals else [])
    outs = {k: [] for k in keys}
    for s in range(0, origins.shape[0], chunk_rays):
        sl = slice(s, s + chunk_rays)
        o, d, l = origins[sl], dirs[sl], lengths[sl]
        # training-mode streams, all injected: stratified depths, density noise, stratified importance samples
        if u_coarse is not None:
            l = jiggle_within_stratas(l, u_coarse[sl])
        dens, col = implicit_function(grid, sd, o, d, l, cfg, prefix)
        rgb_c, dep_c, msk_c, w = ea_raymarch(dens, col, l, cfg, noise_coarse[sl] if noise_coarse is not None else None, noise_std)
        diag = {}
        # (the refiner samples under torch.no_grad() in PyTorch3D's RayPointRefiner: the importance samples carry no gradient)
        lf = refine_lengths(l, w.detach(), cfg, diag, u_fine[sl] if u_fine is not None else None).detach()
        if fine_lengths is not None:
            lf = fine_lengths[sl]
        dens, col = implicit_function(grid, sd, o, d, lf, cfg, prefix)
        rgb, dep, msk, wf = ea_raymarch(dens, col, lf, cfg, noise_fine[sl] if noise_fine is not None else None, noise_std)
        vals = [("rgb", rgb), ("depth", dep), ("mask", msk), ("rgb_c", rgb_c), ("depth_c", dep_c), ("mask_c", msk_c),
                ("fine_lengths", lf), ("pdf_denom", diag["denom"])]
        if with_normals:
            n_c = implicit_normals(grid, sd, o[:, None, :] + l[:, :, None] * d[:, None, :], cfg, prefix)
            n_f = implicit_normals(grid, sd, o[:, None, :] + lf[:, :, None] * d[:, None, :], cfg, prefix)
            vals += [("normals_c", (n_c * w[..., None]).sum(dim=-2)), ("normals", (n_f * wf[..., None]).sum(dim=-2))]
        for k, v in vals:
            outs[k].append(v)
    return {k: torch.cat(v) for k, v in outs.items()}


def render_rays_grad(grid: torch.Tensor, sd: Dict[str, torch.Tensor], origins: torch.Tensor, dirs: torch.Tensor,
                     lengths: torch.Tensor, cfg: RenderCfg, out_grads: Dict[str, torch.Tensor], prefix: str = "", **streams):
    """Backward of ``render_rays`` by autograd (the checker of holo_render_rays_backward): gradients of
    sum_k <out_grads[k], out[k]> over k in rgb / depth / mask / rgb_c / depth_c / mask_c with respect to the grid and
    every RenderMLP parameter - what ``loss.backward()`` leaves on them in the reference for a loss on the renderer's
    outputs (holo_diffusion_model.py:458-489).  Returns (grad_grid, {name: grad}, outputs)."""
    with torch.enable_grad():
        g = grid.detach().clone().requires_grad_(True)
        psd = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        out = render_rays.__wrapped__(g, psd, origins, dirs, lengths, cfg, prefix, **streams)
        loss = sum((out[k] * out_grads[k].reshape(out[k].shape)).sum() for k in out_grads)
        names = [k for k in psd if k.startswith(prefix + "_density_net") or k.startswith(prefix + "_radiance_net")]
        gs = torch.autograd.grad(loss, [g] + [psd[k] for k in names], allow_unused=True)
    pg = {k: (v if v is not None else torch.zeros_like(psd[k])) for k, v in zip(names, gs[1:])}
    return gs[0], pg, {k: v.detach() for k, v in out.items()}


@torch.no_grad()
def render(grid: torch.Tensor, sd: Dict[str, torch.Tensor], cam: dict, cfg: RenderCfg, prefix: str = "",
           chunk_rays: int = 4096, return_coarse: bool = False, with_normals: bool = False) -> Dict[str, torch.Tensor]:
    """Full two-pass render of one camera.  Returns images (1,3,H,W), depths (1,1,H,W), masks (1,1,H,W)."""
    origins, dirs, lengths = make_rays(cam, cfg)
    cat = render_rays(grid, sd, origins, dirs, lengths, cfg, prefix, chunk_rays, with_normals)
    H, W = cfg.image_height, cfg.image_width
    res = {
        "images_render": cat["rgb"].reshape(1, H, W, 3).permute(0, 3, 1, 2).contiguous(),
        "depths_render": cat["depth"].reshape(1, H, W, 1).permute(0, 3, 1, 2).contiguous(),
        "masks_render": cat["mask"].reshape(1, H, W, 1).permute(0, 3, 1, 2).contiguous(),
    }
    if with_normals:
        res["normals_render"] = cat["normals"].reshape(1, H, W, 3).permute(0, 3, 1, 2).contiguous()
        res["normals_coarse"] = cat["normals_c"].reshape(1, H, W, 3).permute(0, 3, 1, 2).contiguous()
    if return_coarse:
        res["images_coarse"] = cat["rgb_c"].reshape(1, H, W, 3).permute(0, 3, 1, 2).contiguous()
        res["depths_coarse"] = cat["depth_c"].reshape(1, H, W, 1).permute(0, 3, 1, 2).contiguous()
        res["masks_coarse"] = cat["mask_c"].reshape(1, H, W, 1).permute(0, 3, 1, 2).contiguous()
        res["fine_lengths"] = cat["fine_lengths"].reshape(H, W, -1)
    return res
