"""Sampling driver: DDPM-sample voxel grids and render fly-arounds, sharded over the GPUs of a node.

Mirrors the callers of the hot path in the reference:
  * ``generate_samples`` (/root/reference/generate_samples.py:37-138): serial loop over
    ``sample_00000..`` sequences, per-sample seeding, ``render_flyaround(sample_mode=True)``
  * ``render_flyaround`` sample-mode branch (holo_diffusion/utils/render_utils/flyaround.py:150-153,
    176-184,219-253): simple-360 trajectory (radius 10, elevation -30 deg, focal 3.2), one
    ``model(**batch, voxel_features=...)`` per camera, optional progressive denoising renders (:240-245)

Multi-GPU: samples are independent DDPM chains, so sample ``i`` goes to rank
``i % world_size`` (one process per GPU, weights replicated); the only communication is one
``all_gather`` of the rendered frames at the end (RCCL over xGMI when the backend is ``nccl``,
``gloo`` in the CPU tests).  Video encoding / visdom / shaded-depth output are out of scope.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .cameras import PerspectiveCameras, get_simple_360_camera_trajectory
from .render import EvaluationMode

CANONICAL_CO3D_UP_AXIS: Tuple[float, float, float] = (-0.0396, -0.8306, -0.5554)  # visualize_reconstruction.py:35


def shard_indices(num_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin assignment of independent samples to ranks."""
    return list(range(rank, num_items, world_size))


def dist_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_frames(local: Dict[int, torch.Tensor], num_items: int, frame_shape: Sequence[int],
                  device: torch.device) -> Optional[torch.Tensor]:
    """all_gather of per-sample frame stacks.  ``local`` maps sample index -> tensor of ``frame_shape``.
    Returns (num_items, *frame_shape) on every rank (None if nothing was rendered)."""
    rank, world = dist_info()
    per_rank = math.ceil(num_items / world) if world > 0 else num_items
    buf = torch.zeros((per_rank,) + tuple(frame_shape), dtype=torch.float32, device=device)
    for slot, idx in enumerate(shard_indices(num_items, rank, world)):
        buf[slot] = local[idx]
    if world == 1:
        return buf[:num_items]
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = torch.zeros((num_items,) + tuple(frame_shape), dtype=torch.float32, device=device)
    for r in range(world):
        for slot, idx in enumerate(shard_indices(num_items, r, world)):
            out[idx] = parts[r][slot]
    return out


@torch.no_grad()
def render_flyaround(*args, **kwargs) -> Dict[str, torch.Tensor]:
    """Sample (unless ``voxel_features`` is given) and render one fly-around.  Returns stacked
    ``images_render (n,3,H,W)``, ``depths_render``, ``masks_render (n,1,H,W)`` and ``voxel_features``.

    Two call forms are accepted:
      * ``render_flyaround(model, n_flyaround_poses=..., ...)`` - the form the drivers of this package use;
      * the reference's signature (flyaround.py:44-80)
        ``render_flyaround(dataset, sequence_name, model, output_video_path, ..., sample_mode=True, ...)``:
        ``dataset`` may be ``None`` in sample mode (flyaround.py:150-153), ``trajectory_type`` must be ``simple_360``
        (the only one that needs no training cameras), displayable frames go to ``output_video_frames_dir`` (or next
        to ``output_video_path``); video encoding / visdom are outside the path.  ``sample_mode=False`` is the
        reconstruction mode of visualize_reconstruction.py:60-162 (see ``_render_flyaround_reconstruction``)."""
    first = args[0] if args else kwargs.get("model", kwargs.get("dataset"))
    if isinstance(first, torch.nn.Module) and "sequence_name" not in kwargs:
        return _render_flyaround(*args, **kwargs)
    return _render_flyaround_reference_form(*args, **kwargs)


def _render_flyaround_reference_form(dataset=None, sequence_name: str = "sample", model=None, output_video_path: str = "",
                                     output_video_name: Optional[str] = None, n_flyaround_poses: int = 40, fps: int = 20,
                                     trajectory_type: str = "circular_lsq_fit", max_angle: float = 2 * math.pi,
                                     trajectory_scale: float = 1.1, scene_center=(0.0, 0.0, 0.0),
                                     up=(0.0, -1.0, 0.0), camera_elevation: float = -30.0 * (2 * math.pi / 360),
                                     camera_focal_length: float = 3.2, hemispherical_radius: float = 10,
                                     traj_offset: float = 0.0, n_source_views: int = 9, visdom_show_preds: bool = False,
                                     visdom_environment: str = "render_flyaround", visdom_server: str = "",
                                     visdom_port: int = 8097, num_workers: int = 10, device="cuda", seed=None,
                                     video_resize=None, output_video_frames_dir: Optional[str] = None,
                                     sample_mode: bool = False, progressive_sampling_steps_per_render: int = -1,
                                     visualize_preds_keys=("images_render", "masks_render", "depths_render"),
                                     save_voxel_features: bool = False) -> Dict[str, torch.Tensor]:
    if model is None:
        raise TypeError("render_flyaround: `model` is required")
    if trajectory_type.lower() != "simple_360":
        raise NotImplementedError(f"trajectory_type '{trajectory_type}' is fitted to training cameras by PyTorch3D's "
                                  "generate_eval_video_cameras; this path supports 'simple_360' (flyaround.py:176-184)")
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if sample_mode:
        out = _render_flyaround(model, n_flyaround_poses=n_flyaround_poses, up=tuple(up), camera_elevation=camera_elevation,
                                camera_focal_length=camera_focal_length, hemispherical_radius=hemispherical_radius,
                                max_angle=max_angle, device=dev,
                                progressive_sampling_steps_per_render=progressive_sampling_steps_per_render)
    else:
        out = _render_flyaround_reconstruction(dataset, sequence_name, model, n_source_views=n_source_views, seed=seed,
                                               n_flyaround_poses=n_flyaround_poses, up=tuple(up),
                                               camera_elevation=camera_elevation, camera_focal_length=camera_focal_length,
                                               hemispherical_radius=hemispherical_radius, max_angle=max_angle, device=dev)
    name = output_video_name or sequence_name
    frames_dir = output_video_frames_dir or (os.path.dirname(output_video_path) if output_video_path else None)
    if frames_dir:
        from .flyaround_output import export_flyaround_frames
        os.makedirs(frames_dir, exist_ok=True)
        export_flyaround_frames({k: v for k, v in out.items() if k in tuple(visualize_preds_keys)}, frames_dir, name)
        if save_voxel_features and out.get("voxel_features") is not None:  # flyaround.py:288-294
            torch.save(out["voxel_features"], os.path.join(frames_dir, f"{sequence_name}_voxel_features.pth"))
    return out


def _frame_field(frame, key):
    return frame.get(key) if isinstance(frame, dict) else getattr(frame, key, None)


def select_source_views(n_frames: int, n_source_views: int, seed) -> List[int]:
    """The reference's reproducible choice of source frames (flyaround.py:164-167): a seeded random permutation of the
    sequence's frames under a forked RNG, the first ``n_source_views`` of it."""
    with torch.random.fork_rng():
        torch.manual_seed(0 if seed is None else int(seed))
        return torch.randperm(n_frames)[:n_source_views].tolist()


def _render_flyaround_reconstruction(dataset, sequence_name: str, model, n_source_views: int = 9, seed=None,
                                     n_flyaround_poses: int = 40, up=(0.0, -1.0, 0.0),
                                     camera_elevation: float = -30.0 * (2 * math.pi / 360), camera_focal_length: float = 3.2,
                                     hemispherical_radius: float = 10, max_angle: float = 2 * math.pi,
                                     device: torch.device = torch.device("cuda")) -> Dict[str, torch.Tensor]:
    """Reconstruction fly-around (flyaround.py:153-171,219-253; visualize_reconstruction.py:60-162): ``n_source_views``
    frames of the sequence are drawn reproducibly, their features are pooled onto the voxel grid, the grid goes through
    ``tanh(net_3d(., 0))`` and is rendered from the fly-around cameras.

    ``dataset``: any object with Implicitron's ``sequence_indices_in_order(sequence_name)`` and ``__getitem__``; a frame
    (attributes or dict keys) carries ``camera`` (``R, T, focal_length, principal_point``), and either ``image_features``
    - the image feature extractor's dict for that frame, key -> (C, H, W) - or ``image_rgb`` (3, H, W) (+ optional
    ``fg_probability``) for ``model.image_feature_extractor``.  The reference re-runs the encoder and the view pooling for
    every rendered frame with identical inputs (only the target camera changes, flyaround.py:222-224); here the grid is
    pooled once and all cameras are rendered in one call - the frames are the same."""
    if dataset is None:
        raise ValueError("render_flyaround(sample_mode=False) needs the dataset of the sequence to reconstruct")
    if not getattr(model, "view_pooler_enabled", False):
        raise ValueError("reconstruction needs a model with view_pooler_enabled")
    seq_idx = list(dataset.sequence_indices_in_order(sequence_name))
    if not seq_idx:
        raise ValueError(f"sequence '{sequence_name}' has no frames")
    frames = [dataset[seq_idx[i]] for i in select_source_views(len(seq_idx), n_source_views, seed)]
    cam_fields = {k: torch.cat([torch.as_tensor(getattr(_frame_field(f, "camera"), k), dtype=torch.float32).reshape(
        (1, 3, 3) if k == "R" else (1, -1)) for f in frames]) for k in ("R", "T", "focal_length", "principal_point")}
    src_cams = PerspectiveCameras(R=cam_fields["R"], T=cam_fields["T"], focal_length=cam_fields["focal_length"],
                                  principal_point=cam_fields["principal_point"]).to(device)
    if all(_frame_field(f, "image_features") is not None for f in frames):
        keys = list(_frame_field(frames[0], "image_features"))
        feats = {k: torch.stack([torch.as_tensor(_frame_field(f, "image_features")[k], dtype=torch.float32) for f in frames]
                                ).to(device) for k in keys}
    else:
        if model.image_feature_extractor is None:
            raise ValueError("the frames carry no `image_features` and the model has no image_feature_extractor "
                             "(PyTorch3D's ResNetFeatureExtractor is outside this path: attach any callable "
                             "(image_rgb, fg_probability) -> {key: (n, C, H, W)})")
        rgb = torch.stack([torch.as_tensor(_frame_field(f, "image_rgb"), dtype=torch.float32) for f in frames]).to(device)
        fg = [_frame_field(f, "fg_probability") for f in frames]
        fg = torch.stack([torch.as_tensor(x, dtype=torch.float32) for x in fg]).to(device) if all(x is not None for x in fg) else None
        feats = model.image_feature_extractor(rgb, fg)
    voxel_features = model.pool_views_to_voxel_features(feats, src_cams)
    cams = get_simple_360_camera_trajectory(max_angle, n_flyaround_poses, camera_elevation, hemispherical_radius, up,
                                            camera_focal_length).to(device)
    out = model.render_views(voxel_features, cams)
    out["voxel_features"] = voxel_features
    return out


def _render_flyaround(model, n_flyaround_poses: int = 40, up: Tuple[float, float, float] = (0.0, -1.0, 0.0),
                      camera_elevation: float = -30.0 * (2 * math.pi / 360), camera_focal_length: float = 3.2,
                      hemispherical_radius: float = 10, max_angle: float = 2 * math.pi,
                      device: torch.device = torch.device("cuda"), progressive_sampling_steps_per_render: int = -1,
                      voxel_features: Optional[torch.Tensor] = None, batched: bool = True,
                      sampler_kwargs: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    cams = get_simple_360_camera_trajectory(max_angle, n_flyaround_poses, camera_elevation, hemispherical_radius, up,
                                            camera_focal_length).to(device)
    sampler_kwargs = dict(sampler_kwargs or {})
    gen = None
    if voxel_features is None and progressive_sampling_steps_per_render <= 0:
        voxel_features = model.sample_random_voxel_features(**sampler_kwargs)
    if progressive_sampling_steps_per_render > 0:
        gen = model.sample_random_voxel_features_progressive(**sampler_kwargs)
    if gen is None and batched:
        out = model.render_views(voxel_features, cams)
        out["voxel_features"] = voxel_features
        return out
    frames = {"images_render": [], "depths_render": [], "masks_render": [], "normals_render": []}
    for n in range(n_flyaround_poses):
        if gen is not None:
            for _ in range(progressive_sampling_steps_per_render):
                try:
                    voxel_features = next(gen)
                except StopIteration:
                    break
        preds = model(camera=cams[n], evaluation_mode=EvaluationMode.EVALUATION, voxel_features=voxel_features)
        for k in frames:
            if k in preds:
                frames[k].append(preds[k])
    out = {k: torch.cat(v, dim=0) for k, v in frames.items() if v}
    out["voxel_features"] = voxel_features
    return out


def _select_cameras(cams, idx: Sequence[int]):
    return cams[list(idx)]


@torch.no_grad()
def render_views_sharded(model, voxel_features: Optional[torch.Tensor], cams, src_rank: int = 0,
                         device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
    """Camera sharding of ONE grid's turntable over the ranks of the process group (last paragraph; the
    per-camera loop of flyaround.py:240-253 has no dependence between cameras): the grid - 33.5 MB at 64^3 x 32, one
    broadcast from the rank that sampled it - goes to every rank, rank r renders cameras r, r + N, r + 2N, ... in one
    ``render_views`` call (the t = 0 refinement runs on every rank: deterministic kernels, bit-equal grids, and the idle
    GPUs cost nothing), and one ``all_gather`` per output puts all frames on all ranks, in camera order.  Bit for bit the
    single-rank ``model.render_views(voxel_features, cams)``: a frame does not depend on which other frames share its
    launch (tests/test_gpu_configs.py::test_teddybear_30_view_turntable_in_one_call).  ``voxel_features`` is read on
    ``src_rank`` only (pass None elsewhere); on a multi-rank run the result also carries the broadcast grid as
    ``"voxel_features"``.  A single-process run is a plain ``render_views``."""
    rank, world = dist_info()
    if world == 1:
        return model.render_views(voxel_features, cams)
    if device is None:
        device = voxel_features.device if voxel_features is not None else torch.device("cuda", torch.cuda.current_device())
    shape = torch.zeros(5, dtype=torch.int64, device=device)
    if rank == src_rank:
        shape.copy_(torch.tensor(voxel_features.shape, dtype=torch.int64))
    dist.broadcast(shape, src=src_rank)
    vf = voxel_features.contiguous().float() if rank == src_rank else torch.empty(tuple(int(v) for v in shape), device=device)
    dist.broadcast(vf, src=src_rank)
    n = len(cams)
    mine = shard_indices(n, rank, world)
    local: Dict[str, Dict[int, torch.Tensor]] = {}
    keys = ["images_render", "depths_render", "masks_render"]
    if mine:
        out = model.render_views(vf, _select_cameras(cams, mine))
        keys = [k for k in out if k.endswith("_render")]
        for k in keys:
            local[k] = {idx: out[k][slot] for slot, idx in enumerate(mine)}
    H, W = model.render_image_height, model.render_image_width
    chans = {"images_render": 3, "normals_render": 3}
    # (every rank must issue the same collectives: the key list of a rank without cameras comes from rank 0's)
    klist = [keys if rank == 0 else None]
    dist.broadcast_object_list(klist, src=0)
    res = {}
    for k in klist[0]:
        res[k] = gather_frames(local.get(k, {}), n, (chans.get(k, 1), H, W), device)
    res["voxel_features"] = vf  # the broadcast grid: every rank holds it
    return res


@torch.no_grad()
def render_progressive_turntable_sharded(model, n_views: int = 30, steps_per_render: int = 1,
                                         up: Tuple[float, float, float] = (0.0, -1.0, 0.0),
                                         camera_elevation: float = -30.0 * (2 * math.pi / 360), camera_focal_length: float = 3.2,
                                         hemispherical_radius: float = 10, max_angle: float = 2 * math.pi,
                                         device: torch.device = torch.device("cuda"), sampler_kwargs: Optional[dict] = None,
                                         src_rank: int = 0):
    """``render_progressive_turntable`` with the cameras of every render sharded over the ranks (configs[3], teddybear.yaml
    - raymarcher-bound: the denoising chain is sequential and stays on ``src_rank``, the 30 views after each step are
    independent).  Every rank iterates this generator; all of them receive all frames; ``voxel_features`` is the
    broadcast grid."""
    rank, world = dist_info()
    cams = get_simple_360_camera_trajectory(max_angle, n_views, camera_elevation, hemispherical_radius, up,
                                            camera_focal_length).to(device)
    gen = model.sample_random_voxel_features_progressive(**dict(sampler_kwargs or {})) if rank == src_rank else None
    while True:
        vf = None
        if rank == src_rank:
            for _ in range(max(1, steps_per_render)):
                try:
                    vf = next(gen)
                except StopIteration:
                    break
        if world > 1:  # the chain's owner tells the others whether another render follows
            flag = torch.tensor([1 if vf is not None else 0], dtype=torch.int64, device=device)
            dist.broadcast(flag, src=src_rank)
            if int(flag.item()) == 0:
                return
        elif vf is None:
            return
        out = render_views_sharded(model, vf, cams, src_rank=src_rank, device=device)
        if world == 1:
            out["voxel_features"] = vf  # (a single-process run is a plain render_views; sharded runs return the broadcast grid)
        yield out


@torch.no_grad()
def render_progressive_turntable(model, n_views: int = 30, steps_per_render: int = 1,
                                 up: Tuple[float, float, float] = (0.0, -1.0, 0.0),
                                 camera_elevation: float = -30.0 * (2 * math.pi / 360), camera_focal_length: float = 3.2,
                                 hemispherical_radius: float = 10, max_angle: float = 2 * math.pi,
                                 device: torch.device = torch.device("cuda"), sampler_kwargs: Optional[dict] = None):
    """BASELINE configs[3] / SURVEY.md 8d config 4: the WHOLE ``n_views`` turntable after every ``steps_per_render``
    denoising steps (progressive semantics of flyaround.py:236-245, with all cameras per step instead of one).  The
    refinement ``tanh(net_3d(vf, 0))`` runs once per step and all cameras go into one batched render call.
    Generator of dicts ``images_render (n_views,3,H,W)``, ``depths_render``, ``masks_render``, ``voxel_features``."""
    cams = get_simple_360_camera_trajectory(max_angle, n_views, camera_elevation, hemispherical_radius, up,
                                            camera_focal_length).to(device)
    gen = model.sample_random_voxel_features_progressive(**dict(sampler_kwargs or {}))
    done = False
    while not done:
        vf = None
        for _ in range(max(1, steps_per_render)):
            try:
                vf = next(gen)
            except StopIteration:
                done = True
                break
        if vf is None:
            break
        out = model.render_views(vf, cams)
        out["voxel_features"] = vf
        yield out


@torch.no_grad()
def generate_samples(model, num_samples: int = 2, n_eval_cameras: int = 25 * 3, seed: int = 3,
                     up: Tuple[float, float, float] = CANONICAL_CO3D_UP_AXIS,
                     camera_elevation: float = -30.0 * (2 * math.pi / 360),
                     progressive_sampling_steps_per_render: int = -1, device: Optional[torch.device] = None,
                     gather: bool = True, sampler_kwargs: Optional[dict] = None,
                     device_noise: bool = False) -> Dict[str, torch.Tensor]:
    """Sharded counterpart of generate_samples.py:105-138.  Every rank renders its own samples; with
    ``gather`` all ranks end up with the frames of all samples.  ``device_noise`` (build-side extension, default off): the
    per-step noise of every chain is drawn inside the step kernel (``ImplicitronGaussianDiffusion.device_noise_seed`` =
    ``seed``, stream = sample index) instead of by ``torch.randn_like``; x_T still comes from the per-sample torch seed."""
    rank, world = dist_info()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    mine = shard_indices(num_samples, rank, world)
    local_img: Dict[int, torch.Tensor] = {}
    local_dep: Dict[int, torch.Tensor] = {}
    local_msk: Dict[int, torch.Tensor] = {}
    H, W = model.render_image_height, model.render_image_width
    diffusion = getattr(model, "diffusion", None)
    perf_noise = device_noise and diffusion is not None and hasattr(diffusion, "device_noise_seed")
    # (the caller's sampler settings come back whatever happens: a later p_sample on this model must not silently stay in
    # the Philox mode with the last sample's stream)
    saved = (diffusion.device_noise_seed, diffusion.device_noise_stream) if perf_noise else None
    try:
        for i in mine:
            torch.manual_seed(seed + i)  # per-sample seed
            if perf_noise:
                diffusion.device_noise_seed, diffusion.device_noise_stream = int(seed), int(i)
            out = render_flyaround(model, n_flyaround_poses=n_eval_cameras, up=up, camera_elevation=camera_elevation,
                                   device=device,
                                   progressive_sampling_steps_per_render=progressive_sampling_steps_per_render,
                                   sampler_kwargs=sampler_kwargs)
            local_img[i], local_dep[i], local_msk[i] = out["images_render"], out["depths_render"], out["masks_render"]
    finally:
        if perf_noise:
            diffusion.device_noise_seed, diffusion.device_noise_stream = saved
    if not gather:
        return {"images_render": local_img, "depths_render": local_dep, "masks_render": local_msk}
    return {
        "images_render": gather_frames(local_img, num_samples, (n_eval_cameras, 3, H, W), device),
        "depths_render": gather_frames(local_dep, num_samples, (n_eval_cameras, 1, H, W), device),
        "masks_render": gather_frames(local_msk, num_samples, (n_eval_cameras, 1, H, W), device),
    }


def generate_samples_from_experiment(exp_dir: str, output_directory: Optional[str] = None,
                                     render_size: Optional[Tuple[int, int]] = None, n_eval_cameras: int = 25 * 3,
                                     num_samples: int = 2, seed: int = 3,
                                     up: Tuple[float, float, float] = CANONICAL_CO3D_UP_AXIS,
                                     camera_elevation: float = -30.0 * (2 * math.pi / 360),
                                     progressive_sampling_steps_per_render: int = -1, save_frames: bool = True,
                                     device: Optional[torch.device] = None, load_fn=None,
                                     device_noise: bool = False) -> Dict[str, torch.Tensor]:
    """``generate_samples(exp_dir=...)`` of the reference script (generate_samples.py:37-138) on the HIP path:
    experiment directory -> model (``checkpoint.load_experiment``) -> sharded sampling + fly-around renders.

    Output stage: rank 0 writes ``<output_directory>/sample_%05d_frames.pt`` (images / depths / masks of the
    fly-around as tensors) and, through ``flyaround_output``, one directory of displayable frames per key; video
    encoding and visdom are outside this path."""
    if load_fn is None:
        from .checkpoint import load_experiment as load_fn  # ``load_fn``: same signature (tests inject a stand-in)
    rank, world = dist_info()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if output_directory is None:
        folder = "generated_samples" if progressive_sampling_steps_per_render == -1 else "generated_samples_denoising"
        output_directory = os.path.join(exp_dir, folder)
    model, report = load_fn(exp_dir, render_size=render_size, device=device)
    if not (model.net_3d_enabled and model.diffusion_enabled):
        raise ValueError("Can generate random samples only from a trained HoloDiffusion model "
                         "(net_3d_enabled and diffusion_enabled)")
    out = generate_samples(model, num_samples=num_samples, n_eval_cameras=n_eval_cameras, seed=seed, up=up,
                           camera_elevation=camera_elevation,
                           progressive_sampling_steps_per_render=progressive_sampling_steps_per_render, device=device,
                           device_noise=device_noise)
    if save_frames and rank == 0:
        from .flyaround_output import export_flyaround_frames
        os.makedirs(output_directory, exist_ok=True)
        for i in range(num_samples):
            torch.save({k: v[i].cpu() for k, v in out.items()},
                       os.path.join(output_directory, f"sample_{i:05d}_frames.pt"))
            # displayable frames per key (flyaround.py:422-488,553-610), ready for a video encoder
            export_flyaround_frames({k: v[i] for k, v in out.items()}, output_directory, f"sample_{i:05d}")
    out["load_report"] = report
    return out


# ---- command line: `python -m holo_diffusion_amd.generate exp_dir=... key=value ...` ------------------------------
# OmegaConf-free counterpart of the reference script's `main()` (generate_samples.py:141-149): the arguments are the
# keyword arguments of `generate_samples` (generate_samples.py:37-51) as `key=value` pairs (values parsed as YAML, like
# OmegaConf.from_cli).  Launched under `torch.distributed.run` it is the multi-GPU product entry of SURVEY.md 8e: one
# process per GPU, `init_process_group("nccl")` (= RCCL over xGMI), `set_device(LOCAL_RANK)`, samples sharded over the
# ranks, one all_gather of the frames at the end.
CLI_DEFAULTS = dict(exp_dir="", output_directory=None, render_size=None, video_size=(256, 256), camera_path="simple_360",
                    n_eval_cameras=25 * 3, num_samples=2, seed=3, trajectory_scale=1.3, up=CANONICAL_CO3D_UP_AXIS,
                    camera_elevation=-30.0 * (2 * math.pi / 360), progressive_sampling_steps_per_render=-1,
                    save_voxel_features=True,
                    device_noise=False)  # (build-side extension: in-kernel Philox noise per denoising step, generate_samples)


def parse_cli(argv: Sequence[str]) -> Dict[str, object]:
    import yaml
    cfg = dict(CLI_DEFAULTS)
    for a in argv:
        if "=" not in a:
            raise SystemExit(f"generate: expected key=value, got '{a}' (keys: {', '.join(CLI_DEFAULTS)})")
        k, v = a.split("=", 1)
        if k not in CLI_DEFAULTS:
            raise SystemExit(f"generate: unknown argument '{k}' (keys: {', '.join(CLI_DEFAULTS)})")
        val = yaml.safe_load(v) if v != "" else ""
        if isinstance(CLI_DEFAULTS[k], tuple) and isinstance(val, list):
            val = tuple(val)
        cfg[k] = val
    if cfg["render_size"] is not None:
        cfg["render_size"] = tuple(int(x) for x in cfg["render_size"])
    return cfg


def init_distributed() -> Tuple[int, int, torch.device]:
    """One process per GPU under torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; backend
    `nccl` (RCCL) on GPUs, `gloo` without (CPU test rigs).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if world > 1 and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if use_gpu:
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    return rank, world, device


def main(argv: Optional[Sequence[str]] = None, load_fn=None) -> int:
    import sys
    cfg = parse_cli(sys.argv[1:] if argv is None else argv)
    if not cfg["exp_dir"]:
        raise SystemExit("generate: exp_dir=<experiment directory> is required")
    if cfg["camera_path"] != "simple_360":
        raise SystemExit("generate: camera_path must be 'simple_360' (the sampling trajectory, flyaround.py:176-184)")
    rank, world, device = init_distributed()
    try:
        with torch.no_grad():
            out = generate_samples_from_experiment(
                cfg["exp_dir"], output_directory=cfg["output_directory"], render_size=cfg["render_size"],
                n_eval_cameras=int(cfg["n_eval_cameras"]), num_samples=int(cfg["num_samples"]), seed=int(cfg["seed"]),
                up=tuple(cfg["up"]), camera_elevation=float(cfg["camera_elevation"]),
                progressive_sampling_steps_per_render=int(cfg["progressive_sampling_steps_per_render"]), device=device,
                load_fn=load_fn, device_noise=bool(cfg["device_noise"]))
        if rank == 0:
            img = out["images_render"]
            print(f"generate: {int(cfg['num_samples'])} samples x {int(cfg['n_eval_cameras'])} frames "
                  f"{tuple(img.shape[-2:])} on {world} rank(s), backend "
                  f"{dist.get_backend() if world > 1 else 'none'}")
    finally:
        if world > 1 and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
