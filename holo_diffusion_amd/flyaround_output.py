"""Fly-around output stage: predictions -> displayable frames.

Follows ``holo_diffusion/utils/render_utils/flyaround.py``:

* ``_images_from_preds`` (:422-488): per key, masks / depths become 3-channel images; depth maps are normalised with
  PyTorch3D's ``make_depth_image`` and composited over a white background with the (nearest-resized) render mask;
* ``_make_shaded_from_normals`` (:400-420) behind the ``_shaded_depth_render`` key (:440-445): a head-light shading
  of the rendered normals (``normals_render``, produced by the fused renderer when the implicit function has
  ``render_normals=True``); the reference's alternative for that key - ``shaded_depth_render.depth_to_shaded``, a
  PyTorch3D mesh/point-cloud rasterisation of the depth map - is outside this path;
* ``_generate_prediction_videos`` (:553-610): one clip per key, frames clipped to [0, 1].

``make_depth_image`` lives in PyTorch3D 0.7.4 (``implicitron/tools/vis_utils.py``), which is not available here: its
published algorithm is restated below (per-image 2 % / 98 % quantiles of the valid masked depths mapped to
[0.1, 0.9]) — PARITY UNPINNED for that function.  Video encoding (ffmpeg via PyTorch3D's VideoWriter) and visdom are
outside this path: frames are written as binary PPM files, one directory per key, ready for any encoder.
Host-side torch code on small frame tensors; nothing here is on the measured hot path.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as Fu


def make_depth_image(depths: torch.Tensor, masks: torch.Tensor, max_quantile: float = 0.98, min_quantile: float = 0.02,
                     min_out_depth: float = 0.1, max_out_depth: float = 0.9) -> torch.Tensor:
    """(N,1,H,W) depths + masks -> (N,1,H,W) in [0,1]: robust per-image normalisation of the foreground depths."""
    normfacs = []
    for d, m in zip(depths, masks):
        ok = (d.reshape(-1) > 1e-6) & (m.reshape(-1) > 0.5)
        if int(ok.sum()) <= 1:
            normfacs.append(torch.zeros(2, dtype=depths.dtype, device=depths.device))
            continue
        dok = d.reshape(-1)[ok]
        maxk = max(int(round((1 - max_quantile) * dok.numel())), 1)
        mink = max(int(round(min_quantile * dok.numel())), 1)
        nmax = dok.topk(k=maxk, dim=-1).values[-1]
        nmin = dok.topk(k=mink, dim=-1, largest=False).values[-1]
        normfacs.append(torch.stack([nmin, nmax]))
    nf = torch.stack(normfacs)
    lo, hi = nf[:, 0].reshape(-1, 1, 1, 1), nf[:, 1].reshape(-1, 1, 1, 1)
    out = (depths - lo) / (hi - lo).clamp(1e-4)
    return ((out * (max_out_depth - min_out_depth) + min_out_depth) * masks.float()).clamp(0.0, 1.0)


def make_shaded_from_normals(n: torch.Tensor, mask: torch.Tensor, diffuse_strength: float = 0.3,
                             specular_strength: float = 0.1, ambient_strength: float = 0.3,
                             specular_hardness: float = 10.0) -> torch.Tensor:
    """``_make_shaded_from_normals`` (flyaround.py:400-420): a point light at the camera centre; the LAST normal
    component is the shading term.  n (N,3,H,W), mask (N or 1,1,H,W) -> (N,1,H,W) in [0,1], white background."""
    shading = n[:, -1:].clamp(0.0)
    shading_ambient = (shading * diffuse_strength + specular_strength * shading ** specular_hardness
                       + ambient_strength) / (diffuse_strength + specular_strength + ambient_strength)
    return (shading_ambient * mask + (1 - mask)).clamp(0.0, 1.0)


def stack_images(ims: torch.Tensor, size=None) -> torch.Tensor:
    """``_stack_images`` (flyaround.py:490-502): a batch of source images tiled into a ceil(sqrt(n))^2 mosaic."""
    ba = ims.shape[0]
    side = int(math.ceil(math.sqrt(ba)))
    n_add = side * side - ba
    if n_add > 0:
        ims = torch.cat((ims, torch.zeros_like(ims[:1]).repeat(n_add, 1, 1, 1)))
    ims = ims.view(side, side, *ims.shape[1:])
    cated = torch.cat([torch.cat(list(row), dim=2) for row in ims], dim=1)
    if size is not None:
        cated = Fu.interpolate(cated[None], size=size, mode="bilinear")[0]
    return cated.clamp(0.0, 1.0)


def images_from_preds(preds: Dict[str, torch.Tensor],
                      extract_keys: Sequence[str] = ("image_rgb", "images_render", "fg_probability", "masks_render",
                                                     "depths_render", "depth_map", "_all_source_images")
                      ) -> Dict[str, torch.Tensor]:
    """``_images_from_preds`` (flyaround.py:422-488, same default keys): every entry becomes an (N,3,H,W) CPU tensor;
    keys the predictions lack are skipped.  ``_all_source_images`` tiles ``image_rgb[1:]`` into one mosaic (:437-439);
    ``_shaded_depth_render`` needs ``normals_render`` (:440-445; the checked-against-the-reference-body fixtures are
    tests/golden/ref_images_from_preds.npz)."""
    imout = {}
    for k in extract_keys:
        if k == "_all_source_images" and preds.get("image_rgb") is not None:
            v = stack_images(preds["image_rgb"][1:].detach().float().cpu().clone(), None)[None]
            imout[k] = v.repeat(1, 3, 1, 1) if v.shape[1] == 1 else v
            continue
        if k == "_shaded_depth_render":
            if preds.get("normals_render") is None:
                continue  # (the depth-map rasterisation fallback of the reference needs PyTorch3D's renderer)
            # (the reference takes the FIRST mask, `preds["masks_render"][:1]`, and broadcasts it: its predictions hold one
            # frame; per-frame shading of a stack of frames is `export_flyaround_frames` below)
            v = make_shaded_from_normals(preds["normals_render"].detach().float().cpu().clone(),
                                         preds["masks_render"][:1].detach().float().cpu().clone())
            imout[k] = v.repeat(1, 3, 1, 1)
            continue
        if k not in preds or preds[k] is None:
            continue
        v = preds[k].detach().float().cpu().clone()
        if k.startswith("depth"):
            mask = Fu.interpolate(preds["masks_render"].detach().float().cpu(), size=v.shape[2:], mode="nearest")
            v = make_depth_image(v, mask)
            v = v * mask + (1 - mask)  # white background
        if v.shape[1] == 1:
            v = v.repeat(1, 3, 1, 1)
        imout[k] = v
    return imout


def write_ppm(path: str, image: torch.Tensor) -> None:
    """(3,H,W) float image in [0,1] -> binary PPM (P6)."""
    img = (image.clamp(0.0, 1.0) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).contiguous()
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(f"P6\n{w} {h}\n255\n".encode())
        f.write(img.numpy().tobytes())


def export_flyaround_frames(frames: Dict[str, torch.Tensor], out_dir: str, sequence_name: str,
                            keys: Optional[Sequence[str]] = None) -> Dict[str, str]:
    """Frames of one fly-around ((F,C,H,W) per key, as ``render_flyaround`` returns them) -> one directory of PPM
    frames per key, named like the reference's per-key clips (``<sequence_name>_<key>``).  Returns key -> directory."""
    default_keys = ("images_render", "masks_render", "depths_render") + (
        ("_shaded_depth_render",) if frames.get("normals_render") is not None else ())
    want = tuple(keys) if keys else default_keys
    ims = images_from_preds(frames, tuple(k for k in want if k != "_shaded_depth_render"))
    if "_shaded_depth_render" in want and frames.get("normals_render") is not None:  # frame by frame: each its own mask
        ims["_shaded_depth_render"] = torch.cat([
            images_from_preds({"normals_render": frames["normals_render"][i:i + 1],
                               "masks_render": frames["masks_render"][i:i + 1]}, ("_shaded_depth_render",))["_shaded_depth_render"]
            for i in range(frames["normals_render"].shape[0])])
    dirs = {}
    for k, v in ims.items():
        d = os.path.join(out_dir, f"{sequence_name}_{k}")
        os.makedirs(d, exist_ok=True)
        for i in range(v.shape[0]):
            write_ppm(os.path.join(d, f"frame_{i:05d}.ppm"), v[i])
        dirs[k] = d
    return dirs
