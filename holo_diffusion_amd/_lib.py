"""ctypes binding of ``libholo_mi355x.so`` (C ABI: ``include/holo_abi.h``).

The shared library is built in-tree by ``make -C holo_diffusion_amd/csrc`` (or
``__graft_entry__.build()``) with ``hipcc --offload-arch=gfx950``.  There is no CPU
fallback: if the library is missing, :func:`load` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libholo_mi355x.so")

HOLO_DTYPE_F32 = 0
HOLO_DTYPE_BF16 = 1
HOLO_DTYPE_F32_BF16X3 = 2
ABI_VERSION = 6  # include/holo_abi.h HOLO_ABI_VERSION


class HoloError(RuntimeError):
    pass


class HoloUnetCfg(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("in_channels", C.c_int32), ("out_channels", C.c_int32),
        ("model_channels", C.c_int32), ("num_res_blocks", C.c_int32),
        ("n_channel_mult", C.c_int32), ("channel_mult", C.c_int32 * 8),
        ("n_attention_resolutions", C.c_int32), ("attention_resolutions", C.c_int32 * 8),
        ("num_heads", C.c_int32), ("homogeneous_resample", C.c_int32),
    ]


class HoloRenderCfg(C.Structure):
    _fields_ = [
        ("resol", C.c_int32), ("feature_size", C.c_int32), ("volume_extent", C.c_float),
        ("scene_extent", C.c_float), ("scene_center", C.c_float * 3),
        ("n_pts_coarse", C.c_int32), ("n_pts_fine", C.c_int32),
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("bg_color", C.c_float * 3), ("background_opacity", C.c_float),
        ("dnet_hidden_dim", C.c_int32), ("dir_emb_dims", C.c_int32), ("sample_pdf_eps", C.c_float),
        ("feature_dim", C.c_int32),
    ]


class HoloCamera(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("T", C.c_float * 3), ("focal", C.c_float * 2),
                ("principal_point", C.c_float * 2)]


class HoloViewFeature(C.Structure):
    _fields_ = [("feats", C.c_void_p), ("channels", C.c_int32), ("height", C.c_int32), ("width", C.c_int32)]


class HoloMlpMeanCfg(C.Structure):
    _fields_ = [("resol", C.c_int32), ("volume_extent", C.c_float), ("feature_size", C.c_int32), ("n_hidden", C.c_int32),
                ("dim_out", C.c_int32), ("n_layers", C.c_int32), ("n_harmonic_functions_ray", C.c_int32),
                ("n_feats", C.c_int32), ("channels", C.c_int32 * 8), ("projection_eps", C.c_float)]


class HoloViewPoolCfg(C.Structure):
    _fields_ = [("resol", C.c_int32), ("volume_extent", C.c_float), ("feature_size", C.c_int32),
                ("weight_by_ray_angle_gamma", C.c_float), ("min_ray_angle_weight", C.c_float),
                ("projection_eps", C.c_float)]


class HoloOpTiming(C.Structure):
    _fields_ = [("op", C.c_int32), ("kernel", C.c_int32), ("tile_depth", C.c_int32), ("fused_skip", C.c_int32),
                ("nsplit", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("out_dim", C.c_int32),
                ("stride", C.c_int32), ("upsample", C.c_int32), ("ksz", C.c_int32), ("ms", C.c_float),
                ("flops", C.c_double), ("flops_executed", C.c_double)]


_vp = C.c_void_p
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); this table is also what tests/test_abi_symbols.py checks against the header
SIGNATURES = {
    "holo_abi_version": (C.c_int, []),
    "holo_last_error": (C.c_char_p, []),
    "holo_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "holo_ctx_destroy": (C.c_int, [_vp]),
    "holo_ctx_set_deterministic": (C.c_int, [_vp, C.c_int]),
    "holo_ctx_get_deterministic": (C.c_int, [_vp]),
    "holo_unet_create": (C.c_int, [_vp, C.POINTER(HoloUnetCfg), C.POINTER(_vp)]),
    "holo_unet_destroy": (C.c_int, [_vp]),
    "holo_unet_num_params": (C.c_int, [_vp]),
    "holo_unet_param_info": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_int, _i64p, C.POINTER(C.c_int)]),
    "holo_unet_set_param": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int, C.c_int, _i64p, _vp]),
    "holo_unet_set_compute_dtype": (C.c_int, [_vp, C.c_int]),
    "holo_unet_workspace_bytes": (C.c_size_t, [_vp, C.c_int]),
    "holo_unet_forward": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_unet_forward_cl": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_unet_fetch_block": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64, _i64p, _vp, _vp]),
    "holo_unet_time_convs": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.c_int, _vp, C.POINTER(C.c_float),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "holo_unet_time_ops": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int, _vp,
                                     C.POINTER(HoloOpTiming), C.c_int, C.POINTER(C.c_int)]),
    "holo_unet_set_dgrad_weight": (C.c_int, [_vp, C.c_char_p, _vp, _vp]),
    "holo_unet_backward_workspace_bytes": (C.c_size_t, [_vp, C.c_int]),
    "holo_unet_backward": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_unet_forward_train": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_unet_backward_taped": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_unet_get_grad": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64, _vp, _vp]),
    "holo_ddpm_step": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int64, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "holo_ddpm_step_philox": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int64, _vp, _vp, C.c_uint64, C.c_uint64, C.c_int,
                                        _vp, _vp, _vp, C.c_int, _vp]),
    "holo_tanh": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "holo_clip": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_float, C.c_int64, _vp]),
    "holo_renderer_create": (C.c_int, [_vp, C.POINTER(HoloRenderCfg), C.POINTER(_vp)]),
    "holo_renderer_destroy": (C.c_int, [_vp]),
    "holo_renderer_set_param": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int, C.c_int, _i64p, _vp]),
    "holo_renderer_commit": (C.c_int, [_vp, _vp]),
    "holo_renderer_set_compute_dtype": (C.c_int, [_vp, C.c_int]),
    "holo_render_workspace_bytes": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "holo_render": (C.c_int, [_vp, _vp, C.POINTER(HoloCamera), C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              C.c_size_t, _vp]),
    "holo_render_rays": (C.c_int, [_vp, _vp, C.POINTER(HoloCamera), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_float,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_render_rays_backward_workspace_bytes": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "holo_render_rays_backward": (C.c_int, [_vp, _vp, C.POINTER(HoloCamera), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_float,
                                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_renderer_get_grad": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64, _vp]),
    "holo_implicit_eval": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_implicit_workspace_bytes": (C.c_size_t, [_vp, C.c_int64, C.c_int64, C.c_int]),
    "holo_implicit_eval_features": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_implicit_normals": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_size_t, _vp]),
    "holo_view_pool_workspace_bytes": (C.c_size_t, [C.POINTER(HoloViewPoolCfg), C.POINTER(HoloViewFeature), C.c_int, C.c_int]),
    "holo_view_pool": (C.c_int, [_vp, C.POINTER(HoloViewPoolCfg), C.POINTER(HoloViewFeature), C.c_int,
                                 C.POINTER(HoloCamera), C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "holo_view_pool_backward_workspace_bytes": (C.c_size_t, [_vp, C.POINTER(HoloViewPoolCfg), C.POINTER(HoloViewFeature),
                                                             C.c_int, C.c_int]),
    "holo_view_pool_backward": (C.c_int, [_vp, C.POINTER(HoloViewPoolCfg), C.POINTER(HoloViewFeature), C.c_int,
                                          C.POINTER(HoloCamera), C.c_int, _vp, _vp, _vp, C.POINTER(_vp), _vp, _vp, _vp,
                                          C.c_size_t, _vp]),
    "holo_mlp_mean_create": (C.c_int, [_vp, C.POINTER(HoloMlpMeanCfg), C.POINTER(_vp)]),
    "holo_mlp_mean_destroy": (C.c_int, [_vp]),
    "holo_mlp_mean_set_param": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int, _i64p, _vp]),
    "holo_mlp_mean_commit": (C.c_int, [_vp, _vp]),
    "holo_mlp_mean_workspace_bytes": (C.c_size_t, [_vp, C.POINTER(HoloViewFeature), C.c_int, C.c_int]),
    "holo_mlp_mean_pool": (C.c_int, [_vp, C.POINTER(HoloViewFeature), C.c_int, C.POINTER(HoloCamera), C.c_int, _vp, _vp,
                                     C.c_size_t, _vp]),
    "holo_mlp_mean_backward_workspace_bytes": (C.c_size_t, [_vp, C.POINTER(HoloViewFeature), C.c_int, C.c_int]),
    "holo_mlp_mean_backward": (C.c_int, [_vp, C.POINTER(HoloViewFeature), C.c_int, C.POINTER(HoloCamera), C.c_int, _vp,
                                         C.POINTER(_vp), _vp, C.c_size_t, _vp]),
    "holo_mlp_mean_get_grad": (C.c_int, [_vp, C.c_char_p, _vp, C.c_int64, _vp]),
    "holo_event_timer_create": (C.c_int, [C.POINTER(_vp)]),
    "holo_event_timer_start": (C.c_int, [_vp, _vp]),
    "holo_event_timer_stop": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "holo_event_timer_destroy": (C.c_int, [_vp]),
}


def bind(cdll: C.CDLL) -> C.CDLL:
    """Attach argtypes/restypes for every exported entry point (raises AttributeError if one is missing)."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


_LIB: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library.  Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HoloError(
                f"{LIB_PATH} not found: the MI355X HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C holo_diffusion_amd/csrc`). "
                "There is no CPU fallback.")
        _LIB = bind(C.CDLL(LIB_PATH))
    return _LIB


def check(lib: C.CDLL, rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.holo_last_error()
        raise HoloError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def shape_array(shape):
    arr = (C.c_int64 * 8)()
    for i, s in enumerate(shape):
        arr[i] = int(s)
    return arr


def make_unet_cfg(image_size, in_channels, out_channels, model_channels, num_res_blocks, channel_mult,
                  attention_resolutions, num_heads, homogeneous_resample=True) -> HoloUnetCfg:
    c = HoloUnetCfg()
    c.image_size, c.in_channels, c.out_channels = int(image_size), int(in_channels), int(out_channels)
    c.model_channels, c.num_res_blocks = int(model_channels), int(num_res_blocks)
    c.n_channel_mult = len(channel_mult)
    for i, v in enumerate(channel_mult):
        c.channel_mult[i] = int(v)
    c.n_attention_resolutions = len(attention_resolutions)
    for i, v in enumerate(attention_resolutions):
        c.attention_resolutions[i] = int(v)
    c.num_heads = int(num_heads)
    c.homogeneous_resample = 1 if homogeneous_resample else 0
    return c


def make_render_cfg(resol, feature_size, image_height, image_width, volume_extent=8.0, scene_extent=4.0,
                    scene_center=(0.0, 0.0, 0.0), n_pts_coarse=64, n_pts_fine=64, bg_color=(1.0, 1.0, 1.0),
                    background_opacity=1e10, dnet_hidden_dim=256, dir_emb_dims=4, sample_pdf_eps=1e-5,
                    feature_dim=0) -> HoloRenderCfg:
    c = HoloRenderCfg()
    c.resol, c.feature_size, c.volume_extent, c.scene_extent = int(resol), int(feature_size), float(volume_extent), float(scene_extent)
    for i in range(3):
        c.scene_center[i] = float(scene_center[i])
        c.bg_color[i] = float(bg_color[i])
    c.n_pts_coarse, c.n_pts_fine = int(n_pts_coarse), int(n_pts_fine)
    c.image_height, c.image_width = int(image_height), int(image_width)
    c.background_opacity = float(background_opacity)
    c.dnet_hidden_dim, c.dir_emb_dims, c.sample_pdf_eps = int(dnet_hidden_dim), int(dir_emb_dims), float(sample_pdf_eps)
    c.feature_dim = int(feature_dim)
    return c
