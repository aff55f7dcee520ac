"""CPU: host-side mirror of the reference plugin interface (no device work)."""
import math
import os

import numpy as np
import pytest
import torch

import holo_diffusion_amd as hda
from holo_diffusion_amd import _lib
from holo_diffusion_amd.structure import unet_param_shapes
from oracle import render_oracle as ro
from oracle import unet_oracle as uo
from oracle.common import NORTH_CFG, PLUMB_CFG, TINY_CFG


@pytest.mark.parametrize("cfg", [TINY_CFG, PLUMB_CFG, NORTH_CFG])
def test_param_layout_matches_reference_names(cfg):
    mine = unet_param_shapes(cfg.image_size, cfg.in_channels, cfg.out_channels, cfg.model_channels,
                             cfg.num_res_blocks, cfg.channel_mult, cfg.attention_resolutions)
    assert mine == uo.unet_param_shapes(cfg)  # oracle map was checked against the reference's state_dict
    if cfg is NORTH_CFG:
        assert len(mine) == 398 and sum(int(np.prod(s)) for s in mine.values()) == 165119392


def test_simple_unet3d_plugin_surface():
    assert hda.registry.get(hda.Unet3DBase, "SimpleUnet3D") is hda.SimpleUnet3D
    d = hda.get_default_args(hda.SimpleUnet3D)
    assert d["image_size"] == 64 and d["channel_mult"] == [1, 2, 4, 8] and d["homogeneous_resample"] is True
    m = hda.SimpleUnet3D(image_size=8, in_channels=32, out_channels=32, model_channels=32, channel_mult=(1, 2),
                         attention_resolutions=(1, 2))
    sd = m.state_dict()
    assert "_net.input_blocks.1.0.in_layers.2.weight" in sd and "_net.out.2.bias" in sd
    assert sd["_net.input_blocks.1.1.proj_out.weight"].abs().max() == 0  # zero_module (unet.py:392)
    assert sd["_net.input_blocks.1.0.in_layers.2.bias"].abs().max() == 0  # diffusion_utils.py:80
    assert next(m.parameters()) is not None
    with pytest.raises(TypeError):
        hda.SimpleUnet3D(not_a_field=1)
    with pytest.raises(_lib.HoloError):  # no CPU fallback
        m(torch.zeros(1, 32, 8, 8, 8), torch.zeros(1, dtype=torch.long))


def test_diffusion_tables_and_indices(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    for T in (1000, 250):
        d = hda.ImplicitronGaussianDiffusion(num_steps=T)
        for k in ("betas", "alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                  "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            assert np.array_equal(getattr(d, k), g[f"T{T}.{k}"]), (T, k)
    d = hda.ImplicitronGaussianDiffusion()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert d._indices(4) == [999, 666, 333, 0]
        assert d._indices(1) == [999]
    assert d._indices(None)[:2] == [999, 998] and len(d._indices(None)) == 1000
    x0 = torch.randn(2, 3, 4)
    t = torch.tensor([0, 999])
    n = torch.randn_like(x0)
    xs = d.q_sample(x0, t, n)
    ref = (torch.from_numpy(d.sqrt_alphas_cumprod)[t].float()[:, None, None] * x0
           + torch.from_numpy(d.sqrt_one_minus_alphas_cumprod)[t].float()[:, None, None] * n)
    torch.testing.assert_close(xs, ref)
    ts, w = d.sample_timesteps(5, torch.device("cpu"))
    assert ts.shape == (5,) and ts.dtype == torch.int64 and torch.all(w == 1)
    with pytest.raises(NotImplementedError):
        hda.ImplicitronGaussianDiffusion(model_mean_type="EPSILON")


def test_cameras_match_oracle():
    for up in ((0.0, -1.0, 0.0), hda.generate.CANONICAL_CO3D_UP_AXIS):
        mine = hda.get_simple_360_camera_trajectory(2 * math.pi, 7, -30.0 * (2 * math.pi / 360), 10, up, 3.2)
        ref = ro.simple_360_cameras(7, up=up)
        torch.testing.assert_close(mine.R, ref["R"])
        torch.testing.assert_close(mine.T, ref["T"])
        torch.testing.assert_close(mine.focal_xy(), ref["focal"])
        torch.testing.assert_close(mine.get_camera_center().norm(dim=1), torch.full((7,), 10.0), rtol=1e-5, atol=1e-5)
    one = mine[3]
    assert len(one) == 1 and one.R.shape == (1, 3, 3)


def test_model_plugin_surface_and_state_dict_names():
    model = hda.HoloDiffusionModel(
        resol=8, feature_size=32, render_image_width=16, render_image_height=8,
        net_3d_SimpleUnet3D_args=dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)),
        diffusion_args=dict(num_steps=250))
    keys = set(model.state_dict().keys())
    assert "net_3d._net.time_embed.0.weight" in keys
    assert "_implicit_functions.0._fn.render_mlp._density_net.mlp.3.0.weight" in keys
    assert "_implicit_functions.1._fn.render_mlp._radiance_net.mlp.0.0.bias" in keys  # shared module, both passes
    assert model.net_3d.in_channels == 32 and model.net_3d.image_size == 8   # holo_diffusion_model.py:121-127
    fn = model._implicit_functions[0]._fn
    assert fn.resol == 8 and fn.n_hidden == 32 and fn.feature_dim == 0        # :152-156
    assert model._implicit_functions[0] is model._implicit_functions[1]       # :165-171
    assert fn.allows_multiple_passes()
    assert model.net_3d_enabled and model.diffusion_enabled and model.n_train_target_views == 10
    assert model.diffusion.num_timesteps == 250
    assert hda.registry.get(hda.model.ImplicitronModelBase, "HoloDiffusionModel") is hda.HoloDiffusionModel
    with pytest.raises(ValueError):
        hda.registry.get(hda.Unet3DBase, "NoSuchNet")


def test_render_mlp_rejects_unsupported_structures():
    with pytest.raises(NotImplementedError):
        hda.RenderMLP(input_dims=32, output_vp_independent_feature_dims=0, dnet_num_layers=3)
    mlp = hda.RenderMLP(input_dims=32, output_vp_independent_feature_dims=0)
    assert {k: tuple(v.shape) for k, v in mlp.state_dict().items()} == ro.render_mlp_param_shapes(
        ro.RenderCfg(feature_size=32))


def test_flyaround_output_stage(tmp_path):
    """_images_from_preds semantics (flyaround.py:422-488): 3-channel outputs, depth normalised inside the mask to
    [0.1, 0.9] and composited over white; frames land as one PPM directory per key."""
    import torch
    from holo_diffusion_amd import flyaround_output as fo
    F, H, W = 3, 6, 8
    g = torch.Generator().manual_seed(0)
    mask = torch.zeros(F, 1, H, W)
    mask[:, :, 1:5, 2:7] = 1.0
    depth = (8.0 + 4.0 * torch.rand(F, 1, H, W, generator=g)) * mask
    preds = {"images_render": torch.rand(F, 3, H, W, generator=g), "masks_render": mask, "depths_render": depth}
    ims = fo.images_from_preds(preds)
    assert set(ims) == {"images_render", "masks_render", "depths_render"}
    for v in ims.values():
        assert v.shape == (F, 3, H, W) and v.min() >= 0.0 and v.max() <= 1.0
    assert torch.equal(ims["masks_render"][:, 0], mask[:, 0])
    d = ims["depths_render"]
    assert torch.all(d[:, :, 0, 0] == 1.0)                      # background composited to white
    inside = d[:, 0][mask[:, 0] > 0.5]
    assert inside.min() >= 0.0 and inside.max() <= 1.0 and 0.1 <= inside.median() <= 0.9
    # monotone in depth inside one frame
    f0 = depth[0, 0][mask[0, 0] > 0.5]
    o0 = d[0, 0][mask[0, 0] > 0.5]
    assert torch.all((o0[f0.argsort()][1:] - o0[f0.argsort()][:-1]) >= -1e-6)
    dirs = fo.export_flyaround_frames(preds, str(tmp_path), "sample_00000")
    import os
    for k, p in dirs.items():
        files = sorted(os.listdir(p))
        assert len(files) == F and files[0] == "frame_00000.ppm"
        head = open(os.path.join(p, files[0]), "rb").read(15)
        assert head.startswith(b"P6\n8 6\n255\n")


def test_camera_host_copy_follows_edits():
    """The host copy the renderer builds its launch parameters from is refreshed when a camera tensor is edited in place
    or reassigned (the cache is keyed on tensor identity + _version)."""
    import holo_diffusion_amd as hda
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 4, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    moved = cams.to("cpu")
    T0 = moved.host()[1].clone()
    moved.T[0, 2] += 1.0
    assert torch.equal(moved.host()[1][0], T0[0] + torch.tensor([0.0, 0.0, 1.0]))
    moved.focal_length = torch.full_like(moved.focal_length, 2.0)
    assert float(moved.host()[2][0, 0]) == 2.0
    sub = moved[[1, 2]]
    sub.R.mul_(-1.0)
    assert torch.equal(sub.host()[0], -moved.host()[0][[1, 2]])


def test_training_entries_validate_their_inputs_and_refuse_the_cpu():
    """Host logic of the training branch: the backward entry needs the forward's draws (no silent fresh draw),
    parameters are created frozen (inference is the default use) and un-freeze like any nn.Module, and - like every entry
    of the package - there is no CPU fallback."""
    import math
    model = hda.HoloDiffusionModel(
        resol=8, feature_size=16, render_image_width=16, render_image_height=16,
        net_3d_SimpleUnet3D_args=dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)))
    assert not any(p.requires_grad for p in model.parameters())
    model.requires_grad_(True)
    assert all(p.requires_grad for p in model.net_3d.parameters())
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 3, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    vf = torch.zeros(1, 16, 8, 8, 8)
    with pytest.raises(ValueError, match="rng_streams"):
        model.training_backward(camera=cams, voxel_features=vf, rng_streams={"timesteps": torch.tensor([5])}, grads={})
    rs = {"timesteps": torch.tensor([5]), "q_noise": torch.zeros_like(vf), "bootstrap": False, "xys": torch.zeros(2, 4, 2)}
    with pytest.raises(hda._lib.HoloError, match="no CPU fallback"):
        model.training_backward(camera=cams, voxel_features=vf, rng_streams=rs, grads={})
