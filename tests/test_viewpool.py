"""View pooling: known-answer tests of the CPU restatement (PARITY UNPINNED: the arithmetic is
PyTorch3D's, see oracle/viewpool_oracle.py) and, with `-m gpu`, the fused HIP kernel against it."""
import math
import os

import numpy as np
import pytest
import torch

import holo_diffusion_amd as hda
from holo_diffusion_amd.weights import synth_state_dict
from oracle import render_oracle as ro
from oracle import viewpool_oracle as vo
from oracle.common import np_noise


def _cams(n, radius=10.0):
    return ro.simple_360_cameras(n, radius=radius)


def test_coord_grid_is_the_renderers_voxel_centres():
    """Trilinear fetch (the renderer's locator) at the pooled points returns the voxels themselves."""
    R, C = 6, 5
    grid = torch.from_numpy(np_noise(1, (1, C, R, R, R)))
    pts = vo.coord_grid(R, 8.0)
    f = ro.trilinear(grid, pts, ro.RenderCfg(resol=R, feature_size=C))
    ref = grid[0].reshape(C, -1).t()
    assert (f - ref).abs().max() < 1e-5
    assert torch.allclose(pts[0], torch.full((3,), -0.5 * (R - 1) * (8.0 / R)))


def test_points_on_a_pixel_ray_project_to_that_pixel_and_sample_it():
    """project_ndc inverts the ray sampler's un-projection; ndc_grid_sample at a pixel-centre NDC returns the pixel."""
    H, W = 6, 10  # non-square: the longer side spans +-(W/H) in NDC
    rcfg = ro.RenderCfg(image_height=H, image_width=W)
    cams = _cams(4)
    cam = {k: v[1:2] for k, v in cams.items()}
    o, d, l = ro.make_rays(cam, rcfg)
    xy = ro.ndc_pixel_grid(H, W).reshape(-1, 2)
    for depth in (7.0, 10.0, 12.5):
        ndc = vo.project_ndc(o + depth * d, cams["R"][1], cams["T"][1], cams["focal"][1], cams["pp"][1])
        assert (ndc - xy).abs().max() < 2e-5
    img = torch.from_numpy(np_noise(3, (4, H, W)))
    s = vo.ndc_grid_sample(img, xy)
    assert (s - img.reshape(4, -1).t()).abs().max() < 1e-5


def test_aggregator_closed_forms():
    R = 4
    cams = _cams(3)
    const = {"a": torch.full((3, 2, 5, 5), 0.7)}
    agg = vo.pool_views(const, cams, R, 0.5)  # a tiny volume at the centre: every voxel projects inside every view
    assert agg.shape == (R ** 3, 4)
    assert (agg[:, :2] - 0.7).abs().max() < 1e-6           # AVG of identical views
    assert (agg[:, 2:] - 1e-2).abs().max() < 1e-6          # STD = sqrt(clamp(0, 1e-4))
    # two opposite cameras: the second view's direction is the negated first -> angular weight clamps at min
    R0, T0 = ro.look_at_view_transform(10.0, 0.0, 0.0)
    R1, T1 = ro.look_at_view_transform(10.0, 0.0, 180.0)
    two = {"R": torch.cat([R0, R1]), "T": torch.cat([T0, T1]), "focal": torch.full((2, 2), 3.2), "pp": torch.zeros(2, 2)}
    f = {"a": torch.stack([torch.zeros(1, 4, 4), torch.ones(1, 4, 4)])}
    agg = vo.pool_views(f, two, 2, 0.05, min_weight=0.1)
    # weights ~ (1, 0.1): AVG = 0.1 / 1.1
    assert (agg[:, 0] - 0.1 / 1.1).abs().max() < 2e-3


# --------------------------------------------------------------------------------------------------------------
def _synthetic_views(n_src, seed, coarse=False):
    """coarse: maps of a few pixels - most voxels of a line (and, on an 8^3 grid, of the next line) share a bilinear cell."""
    s0, s1, s2 = ((4, 6), (5, 5), (3, 4)) if coarse else ((20, 24), (30, 30), (17, 13))
    feats = {"res": torch.tanh(torch.from_numpy(np_noise(seed, (n_src, 16) + s0))),
             "mask": torch.sigmoid(torch.from_numpy(np_noise(seed + 1, (n_src, 1) + s1))),
             "rgb": torch.sigmoid(torch.from_numpy(np_noise(seed + 2, (n_src, 3) + s2)))}
    A = 2 * (16 + 1 + 3)
    return feats, A


@pytest.mark.gpu
@pytest.mark.parametrize("R,n_src,radius,gamma", [(8, 3, 10.0, 1.0), (16, 5, 6.0, 2.0), (8, 9, 3.0, 1.0)])
def test_view_pool_kernel_vs_oracle(R, n_src, radius, gamma):
    """holo_view_pool (projection + bilinear gather + angle-weighted AVG/STD + mapper + tanh, one kernel) against the
    oracle: non-square maps of three sizes, channel counts 16 / 1 / 3, cameras far, near and INSIDE the volume's
    bounding sphere (radius 3: many voxels project outside the images or behind a camera -> zeros padding, |z| clamp)."""
    import tests.gpu_utils as gu
    F = 16
    feats, A = _synthetic_views(n_src, 50 + R)
    cams_d = _cams(n_src, radius=radius)
    w = synth_state_dict({"w": (F, A), "b": (F,)}, 9)
    w["b"] = 0.1 * torch.from_numpy(np_noise(4, (F,)))
    ref = vo.voxel_features_from_views(feats, cams_d, w["w"], w["b"], R, 8.0, gamma=gamma)
    model = hda.HoloDiffusionModel(resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False,
                                   diffusion_enabled=False, render_image_width=8, render_image_height=8,
                                   view_pooler_args=dict(feature_aggregator_AngleWeightedReductionFeatureAggregator_args=dict(
                                       weight_by_ray_angle_gamma=gamma)))
    model.load_state_dict({"pooled_feature_mapper.weight": w["w"], "pooled_feature_mapper.bias": w["b"]}, strict=False)
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    got = model.pool_views_to_voxel_features({k: v.to(gu.DEV) for k, v in feats.items()}, cams.to(gu.DEV))
    assert got.shape == (1, F, R, R, R)
    assert (got.cpu() - ref).abs().max().item() < 1e-4
    assert ref.abs().max() <= 1.0 and ref.std() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("R,n_src,radius,gamma", [(8, 3, 10.0, 1.0), (16, 5, 6.0, 2.0), (8, 9, 3.0, 1.0), (6, 2, 7.0, 1.0)])
def test_view_pool_backward_vs_autograd_of_the_oracle(R, n_src, radius, gamma):
    """holo_view_pool_backward against torch autograd through the oracle's forward (the reference trains this branch with
    autograd, holo_diffusion_model.py:340-373): gradients of the three feature maps, of pooled_feature_mapper.weight and
    .bias for a random cotangent on the grid.  Same geometries as the forward test (cameras inside the volume: taps with
    zero weight, |z| clamp; 6^3 = 216 voxels: the last group of 16 is half empty and its rows wrap over three lines).  STD = sqrt(clamp(var, 1e-4)): a channel whose variance sits within rounding of the clamp
    takes the other branch on one side - the maps here keep var far from 1e-4 except for voxels that project outside
    every view (all samples zero, var = 0: clamped, zero gradient, on both sides)."""
    import tests.gpu_utils as gu
    F = 16
    feats, A = _synthetic_views(n_src, 50 + R)
    cams_d = _cams(n_src, radius=radius)
    w = synth_state_dict({"w": (F, A), "b": (F,)}, 9)
    w["b"] = 0.1 * torch.from_numpy(np_noise(4, (F,)))
    g = torch.from_numpy(np_noise(123, (1, F, R, R, R)))
    leaves = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    mw, mb = w["w"].clone().requires_grad_(True), w["b"].clone().requires_grad_(True)
    out = vo.voxel_features_from_views(leaves, cams_d, mw, mb, R, 8.0, gamma=gamma)
    out.backward(g)
    model = hda.HoloDiffusionModel(resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False,
                                   diffusion_enabled=False, render_image_width=8, render_image_height=8,
                                   view_pooler_args=dict(feature_aggregator_AngleWeightedReductionFeatureAggregator_args=dict(
                                       weight_by_ray_angle_gamma=gamma)))
    model.load_state_dict({"pooled_feature_mapper.weight": w["w"], "pooled_feature_mapper.bias": w["b"]}, strict=False)
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    dev_feats = {k: v.to(gu.DEV) for k, v in feats.items()}
    model.pool_views_to_voxel_features(dev_feats, cams.to(gu.DEV))
    got = model.pool_views_backward(dev_feats, cams.to(gu.DEV), g.to(gu.DEV))

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(got["pooled_feature_mapper"]["weight"], mw.grad) < 1e-3, rel(got["pooled_feature_mapper"]["weight"], mw.grad)
    assert rel(got["pooled_feature_mapper"]["bias"], mb.grad) < 1e-3
    for k in feats:
        assert got["image_features"][k].shape == feats[k].shape
        assert float(leaves[k].grad.abs().max()) > 0
        assert rel(got["image_features"][k], leaves[k].grad) < 1e-3, (k, rel(got["image_features"][k], leaves[k].grad))
    # the mapper-only form (no feature-map gradients requested) gives the same parameter gradients
    got2 = model.pool_views_backward(dev_feats, cams.to(gu.DEV), g.to(gu.DEV), want_feature_grads=False)
    assert got2["image_features"] == {} and torch.equal(got2["pooled_feature_mapper"]["weight"], got["pooled_feature_mapper"]["weight"])


@pytest.mark.gpu
def test_view_pool_backward_released_channel_structure(monkeypatch):
    """The released configuration's channel structure (configs/apple.yaml:166-196: four 16-channel ResNet stages + mask + image =
    18 channel quads, 136 aggregated features) at small map sizes: the scatter's rows walk the quads in TWO rounds (16 + 2), the
    second with fourteen idle rows, and most voxels of a row share a bilinear cell on the coarse maps (the segmented sums, the
    staged whole-pixel atomics) - against autograd through the oracle, and the first-form kernel (HOLO_VIEWPOOL_BWD_V1=1)
    against the same reference."""
    import tests.gpu_utils as gu
    R, n_src, F = 8, 4, 32
    feats = {f"res{i}": torch.tanh(torch.from_numpy(np_noise(70 + i, (n_src, 16, s, s + 2)))) for i, s in enumerate((12, 9, 5, 3))}
    feats["mask"] = torch.sigmoid(torch.from_numpy(np_noise(75, (n_src, 1, 20, 20))))
    feats["rgb"] = torch.sigmoid(torch.from_numpy(np_noise(76, (n_src, 3, 20, 20))))
    A = 2 * sum(v.shape[1] for v in feats.values())
    assert A == 136
    cams_d = _cams(n_src, radius=6.0)
    w = synth_state_dict({"w": (F, A), "b": (F,)}, 9)
    w["b"] = 0.1 * torch.from_numpy(np_noise(4, (F,)))
    g = torch.from_numpy(np_noise(123, (1, F, R, R, R)))
    leaves = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    mw, mb = w["w"].clone().requires_grad_(True), w["b"].clone().requires_grad_(True)
    vo.voxel_features_from_views(leaves, cams_d, mw, mb, R, 8.0).backward(g)
    model = hda.HoloDiffusionModel(resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False,
                                   diffusion_enabled=False, render_image_width=8, render_image_height=8)
    model.load_state_dict({"pooled_feature_mapper.weight": w["w"], "pooled_feature_mapper.bias": w["b"]}, strict=False)
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    dev_feats = {k: v.to(gu.DEV) for k, v in feats.items()}

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12))
    for v1 in ("0", "1"):
        monkeypatch.setenv("HOLO_VIEWPOOL_BWD_V1", v1)
        got = model.pool_views_backward(dev_feats, cams.to(gu.DEV), g.to(gu.DEV))
        assert rel(got["pooled_feature_mapper"]["weight"], mw.grad) < 1e-3, v1
        assert rel(got["pooled_feature_mapper"]["bias"], mb.grad) < 1e-3, v1
        for k in feats:
            assert float(leaves[k].grad.abs().max()) > 0
            assert rel(got["image_features"][k], leaves[k].grad) < 1e-3, (v1, k, rel(got["image_features"][k], leaves[k].grad))


@pytest.mark.gpu
@pytest.mark.parametrize("v1", ["0", "1"])
def test_view_pool_backward_deterministic_mode(v1, monkeypatch):
    """holo_ctx_set_deterministic on holo_view_pool_backward: the bilinear scatter-add of the feature-map gradients as 64-bit
    fixed-point sums (one measuring launch for the binary point of every map, one adding launch).  Two calls are bit-identical,
    the result sits within fp32 summation noise of the default mode's (hardware fp32 atomics), the mapper's gradients - fixed-
    order sums in both modes - do not change at all; both kernel forms (HOLO_VIEWPOOL_BWD_V1)."""
    import tests.gpu_utils as gu
    from holo_diffusion_amd import runtime
    R, n_src, F = 8, 4, 32
    feats = {f"res{i}": torch.tanh(torch.from_numpy(np_noise(70 + i, (n_src, 16, s, s + 2)))) for i, s in enumerate((12, 9, 5, 3))}
    feats["mask"] = torch.sigmoid(torch.from_numpy(np_noise(75, (n_src, 1, 20, 20))))
    feats["rgb"] = torch.sigmoid(torch.from_numpy(np_noise(76, (n_src, 3, 20, 20))))
    A = 2 * sum(v.shape[1] for v in feats.values())
    cams_d = _cams(n_src, radius=6.0)
    w = synth_state_dict({"w": (F, A), "b": (F,)}, 9)
    g = torch.from_numpy(np_noise(123, (1, F, R, R, R)))
    model = hda.HoloDiffusionModel(resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False,
                                   diffusion_enabled=False, render_image_width=8, render_image_height=8)
    model.load_state_dict({"pooled_feature_mapper.weight": w["w"], "pooled_feature_mapper.bias": w["b"]}, strict=False)
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    dev_feats = {k: v.to(gu.DEV) for k, v in feats.items()}
    monkeypatch.setenv("HOLO_VIEWPOOL_BWD_V1", v1)

    def run():
        out = model.pool_views_backward(dev_feats, cams.to(gu.DEV), g.to(gu.DEV))
        return ({k: v.clone() for k, v in out["image_features"].items()},
                {k: v.clone() for k, v in out["pooled_feature_mapper"].items()})

    with runtime.deterministic(True, gu.DEV):
        f1, m1 = run()
        f2, m2 = run()
    with runtime.deterministic(False, gu.DEV):
        fa, ma = run()
    for k in m1:
        assert torch.equal(m1[k], m2[k]) and torch.equal(m1[k], ma[k]), k
    for k in feats:
        scale = float(fa[k].abs().max())
        assert scale > 0 and torch.equal(f1[k], f2[k]), k
        assert float((f1[k] - fa[k]).abs().max()) < 2e-5 * scale, (k, float((f1[k] - fa[k]).abs().max()) / scale)


@pytest.mark.gpu
def test_model_forward_from_source_views():
    """HoloDiffusionModel.forward with source-view features (the reconstruction entry, holo_diffusion_model.py:327-374):
    camera batch = [target, sources...]; the pooled grid goes through tanh(net_3d(., 0)) and the renderer like a
    sampled one.  The frame equals rendering the pooled grid directly."""
    import tests.gpu_utils as gu
    n_src, R, F = 4, 8, 16
    feats, A = _synthetic_views(n_src, 77)
    model, *_ = gu.make_model(R, F, 10, 12, dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)))
    assert model.view_pooler is None
    m2 = hda.HoloDiffusionModel(resol=R, feature_size=F, render_image_width=12, render_image_height=10,
                                view_pooler_enabled=True,
                                net_3d_SimpleUnet3D_args=dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)))
    sd = {k: v for k, v in model.state_dict().items()}
    w = synth_state_dict({"pooled_feature_mapper.weight": (F, A), "pooled_feature_mapper.bias": (F,)}, 5)
    m2.load_state_dict({**{k: v.cpu() for k, v in sd.items()}, **w})
    m2.to(gu.DEV)
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, n_src + 1, -0.5, 10, (0.0, -1.0, 0.0), 3.2).to(gu.DEV)
    dfeats = {k: v.to(gu.DEV) for k, v in feats.items()}
    preds = m2(camera=cams, image_features=dfeats)
    assert preds["images_render"].shape == (1, 3, 10, 12) and torch.isfinite(preds["images_render"]).all()
    vf = m2.pool_views_to_voxel_features(dfeats, cams[list(range(1, n_src + 1))])
    direct = m2(camera=cams[0], voxel_features=vf)
    assert torch.equal(direct["images_render"], preds["images_render"])
    # an extractor callable is used when raw images come in
    m2.image_feature_extractor = lambda rgb, fg: dfeats
    p3 = m2(camera=cams, image_rgb=torch.zeros(n_src + 1, 3, 8, 8, device=gu.DEV))
    assert torch.equal(p3["images_render"], preds["images_render"])
    # a mixed-sequence batch (safe_slice_sources, holo_diffusion_model.py:276-298): only the frames of the FIRST frame's
    # sequence, minus the target, are pooled
    names = ["seq_a", "seq_a", "seq_b", "seq_a", "seq_b"]
    feats5, _ = _synthetic_views(n_src + 1, 91)
    d5 = {k: v.to(gu.DEV) for k, v in feats5.items()}
    p4 = m2(camera=cams, image_features=d5, sequence_name=names)
    vf4 = m2.pool_views_to_voxel_features({k: v[[1, 3]] for k, v in d5.items()}, cams[[1, 3]])
    assert torch.equal(m2(camera=cams[0], voxel_features=vf4)["images_render"], p4["images_render"])
    p5 = m2(camera=cams, image_features=d5)  # no names: every frame after the target is a source
    assert not torch.equal(p5["images_render"], p4["images_render"])


# ---- MLPMeanFeatureAggregator (configs/hydrant.yaml:184) ------------------------------------------------------------
def test_mlp_mean_config_and_state_dict_names():
    """The plugin carries the reference's config fields and state-dict names (custom_modules.py:171-196), the LazyLinear
    layers take their width from a checkpoint, and an experiment config that selects the aggregator keeps view pooling."""
    from holo_diffusion_amd.viewpool import MLPMeanFeatureAggregator, ViewPooler
    vp = ViewPooler(feature_aggregator_class_type="MLPMeanFeatureAggregator",
                    feature_aggregator_MLPMeanFeatureAggregator_args=dict(n_hidden=128, dim_out=128, n_layers=1,
                                                                          n_harmonic_functions_ray=3, checkpointed_mlp=True))
    agg = vp.feature_aggregator
    assert isinstance(agg, MLPMeanFeatureAggregator) and agg.get_aggregated_feature_dim({}) == 128
    shapes = vo.mlp_mean_param_shapes(89)
    assert set(agg.state_dict()) == set(shapes)
    sd = synth_state_dict(shapes, 5)
    agg.load_state_dict(sd)
    assert tuple(agg._first_sampled.weight.shape) == (128, 89) and torch.equal(agg._mlp.mlp[0][0].weight, sd["_mlp.mlp.0.0.weight"])
    from tests.test_checkpoint_loading import _expconfig
    from holo_diffusion_amd import checkpoint as ck
    cfg = _expconfig()
    margs = cfg["model_factory_ImplicitronModelFactory_args"]["model_HoloDiffusionModel_args"]
    margs["view_pooler_args"] = {"feature_aggregator_class_type": "MLPMeanFeatureAggregator",
                                 "view_sampler_args": {"masked_sampling": False, "sampling_mode": "bilinear"},
                                 "feature_aggregator_MLPMeanFeatureAggregator_args": {
                                     "exclude_target_view": True, "exclude_target_view_mask_features": True,
                                     "concatenate_output": True, "n_hidden": 128, "dim_out": 128, "n_layers": 1,
                                     "n_harmonic_functions_ray": 3, "checkpointed_mlp": True}}
    kw, ignored = ck.model_args_from_expconfig(cfg)
    assert kw["view_pooler_enabled"] is True and not any("view pooling disabled" in s for s in ignored)
    model = hda.HoloDiffusionModel(**kw)
    assert isinstance(model.view_pooler.feature_aggregator, MLPMeanFeatureAggregator)
    assert model.view_pooler.feature_aggregator.exclude_target_view is False  # forced off (holo_diffusion_model.py:114-116)
    assert any(k.startswith("view_pooler.feature_aggregator._first_sampled.") for k in model.state_dict())


@pytest.mark.gpu
@pytest.mark.parametrize("R,n_src,radius,F,dim_out,n_harm", [(8, 3, 10.0, 16, 24, 3), (16, 5, 6.0, 32, 128, 3), (8, 9, 3.0, 16, 128, 2)])
def test_mlp_mean_view_pool_kernel_vs_oracle(R, n_src, radius, F, dim_out, n_harm):
    """holo_mlp_mean_pool (projection + bilinear gather + ray-direction embedding + the folded MLPMeanFeatureAggregator on
    the matrix cores + softmax over views + mapper + tanh, one kernel) against the oracle, whose aggregator is bit-equal
    to the reference class (tests/golden/ref_mlp_mean_aggregator.npz); cameras far, near and INSIDE the bounding sphere."""
    import tests.gpu_utils as gu
    feats, _ = _synthetic_views(n_src, 150 + R)
    D = 16 + 1 + 3 + 3 * (2 * n_harm + 1)
    cams_d = _cams(n_src, radius=radius)
    shapes = vo.mlp_mean_param_shapes(D, 128, dim_out)
    sd = synth_state_dict(shapes, 21)
    for k in shapes:
        if k.endswith("bias"):
            sd[k] = 0.1 * torch.from_numpy(np_noise(len(k), shapes[k]))
    w = synth_state_dict({"w": (F, dim_out), "b": (F,)}, 9)
    w["b"] = 0.1 * torch.from_numpy(np_noise(4, (F,)))
    ref = vo.voxel_features_from_views_mlp_mean(feats, cams_d, sd, w["w"], w["b"], R, 8.0, n_harmonic=n_harm)
    model = hda.HoloDiffusionModel(
        resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False, diffusion_enabled=False,
        render_image_width=8, render_image_height=8,
        view_pooler_args=dict(feature_aggregator_class_type="MLPMeanFeatureAggregator",
                              feature_aggregator_MLPMeanFeatureAggregator_args=dict(dim_out=dim_out, n_harmonic_functions_ray=n_harm)))
    full = {"pooled_feature_mapper.weight": w["w"], "pooled_feature_mapper.bias": w["b"]}
    full.update({"view_pooler.feature_aggregator." + k: v for k, v in sd.items()})
    res = model.load_state_dict(full, strict=False)
    assert not res.unexpected_keys
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    got = model.pool_views_to_voxel_features({k: v.to(gu.DEV) for k, v in feats.items()}, cams.to(gu.DEV))
    assert got.shape == (1, F, R, R, R)
    assert (got.cpu() - ref).abs().max().item() < 1e-4
    assert ref.abs().max() <= 1.0 and ref.std() > 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("R,n_src,radius,F,dim_out,n_harm,coarse", [(8, 3, 10.0, 16, 24, 3, False), (16, 5, 6.0, 32, 128, 3, False),
                                                                     (8, 9, 3.0, 16, 128, 2, False), (8, 3, 8.0, 16, 24, 3, True)])
def test_mlp_mean_view_pool_backward_vs_autograd_of_the_oracle(R, n_src, radius, F, dim_out, n_harm, coarse):
    """holo_mlp_mean_backward against torch autograd through the oracle (whose aggregator is bit-equal to the reference
    class): gradients of every aggregator parameter, of pooled_feature_mapper and of the three source-view feature maps for a
    random cotangent on the grid; the geometries of the forward test.  LeakyReLU kink: a hidden pre-activation within float32
    rounding of zero takes the other slope on one side - one unit of one (voxel, view) row, far below the tolerance."""
    import tests.gpu_utils as gu
    feats, _ = _synthetic_views(n_src, 150 + R, coarse=coarse)  # (coarse: the scatter's runs of voxels in one cell, wrapped rows)
    D = 16 + 1 + 3 + 3 * (2 * n_harm + 1)
    cams_d = _cams(n_src, radius=radius)
    shapes = vo.mlp_mean_param_shapes(D, 128, dim_out)
    sd = synth_state_dict(shapes, 21)
    for k in shapes:
        if k.endswith("bias"):
            sd[k] = 0.1 * torch.from_numpy(np_noise(len(k), shapes[k]))
    w = synth_state_dict({"w": (F, dim_out), "b": (F,)}, 9)
    w["b"] = 0.1 * torch.from_numpy(np_noise(4, (F,)))
    g = torch.from_numpy(np_noise(321, (1, F, R, R, R)))
    leaves = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mw, mb = w["w"].clone().requires_grad_(True), w["b"].clone().requires_grad_(True)
    vo.voxel_features_from_views_mlp_mean(leaves, cams_d, psd, mw, mb, R, 8.0, n_harmonic=n_harm).backward(g)
    model = hda.HoloDiffusionModel(
        resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False, diffusion_enabled=False,
        render_image_width=8, render_image_height=8,
        view_pooler_args=dict(feature_aggregator_class_type="MLPMeanFeatureAggregator",
                              feature_aggregator_MLPMeanFeatureAggregator_args=dict(dim_out=dim_out, n_harmonic_functions_ray=n_harm)))
    full = {"pooled_feature_mapper.weight": w["w"], "pooled_feature_mapper.bias": w["b"]}
    full.update({"view_pooler.feature_aggregator." + k: v for k, v in sd.items()})
    model.load_state_dict(full, strict=False)
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    dev_feats = {k: v.to(gu.DEV) for k, v in feats.items()}
    model.pool_views_to_voxel_features(dev_feats, cams.to(gu.DEV))
    got = model.pool_views_backward(dev_feats, cams.to(gu.DEV), g.to(gu.DEV))

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(got["pooled_feature_mapper"]["weight"], mw.grad) < 1e-3, rel(got["pooled_feature_mapper"]["weight"], mw.grad)
    assert rel(got["pooled_feature_mapper"]["bias"], mb.grad) < 1e-3
    assert set(got["feature_aggregator"]) == set(shapes)
    for k in shapes:
        assert float(psd[k].grad.abs().max()) > 0
        assert rel(got["feature_aggregator"][k], psd[k].grad) < 1e-3, (k, rel(got["feature_aggregator"][k], psd[k].grad))
    for k in feats:
        assert rel(got["image_features"][k], leaves[k].grad) < 1e-3, (k, rel(got["image_features"][k], leaves[k].grad))


def _ref_mlp_mean_backward_fixture():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_mlp_mean_backward.npz"))
    R, n_src, F, dim_out, n_hidden, n_harm = (int(v) for v in g["dims"])
    pick = lambda pre: {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}  # noqa: E731
    return g, (R, n_src, F, dim_out, n_hidden, n_harm), pick


def test_oracle_mlp_mean_backward_vs_reference_gradients():
    """tests/golden/ref_mlp_mean_backward.npz: gradients of the REFERENCE's MLPMeanFeatureAggregator (custom_modules.py:162-293,
    AST-executed by oracle/make_golden_render.py) + pooled_feature_mapper + tanh under torch autograd.  The oracle's autograd
    on the same inputs reproduces them (bit-equal when the fixture was written: 1e-6 here, another host's reductions), which
    pins the checker of holo_mlp_mean_backward to the reference class.  The fixture also records the reference's
    checkpoint quirk (checkpointed_mlp = True detaches the aggregate under the re-entrant checkpoint of torch 1.13.1)."""
    g, (R, n_src, F, dim_out, n_hidden, n_harm), pick = _ref_mlp_mean_backward_fixture()
    assert int(g["checkpointed_output_is_detached"]) == 1
    cams = pick("cam.")
    leaves = {k: v.clone().requires_grad_(True) for k, v in pick("maps.").items()}
    psd = {k: v.clone().requires_grad_(True) for k, v in pick("param.").items()}
    mw, mb = pick("mapper.")["weight"].clone().requires_grad_(True), pick("mapper.")["bias"].clone().requires_grad_(True)
    out = vo.voxel_features_from_views_mlp_mean(leaves, cams, psd, mw, mb, R, float(g["volume_extent"]), n_harmonic=n_harm)
    assert (out.detach() - torch.from_numpy(g["out"])).abs().max() < 2e-6
    out.backward(torch.from_numpy(g["cot"]))
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))  # noqa: E731
    for k, r in pick("grad.param.").items():
        assert rel(psd[k].grad, r) < 1e-5, k
    for k, r in pick("grad.maps.").items():
        assert rel(leaves[k].grad, r) < 1e-5, k
    assert rel(mw.grad, pick("grad.mapper.")["weight"]) < 1e-5 and rel(mb.grad, pick("grad.mapper.")["bias"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [None, "128"])
def test_mlp_mean_view_pool_backward_vs_reference_gradients(chunk, monkeypatch):
    """holo_mlp_mean_backward against the gradients of the reference class itself (the fixture above): every aggregator
    parameter, the mapper and the three source-view feature maps at 1e-3 of each tensor's scale; forward at 1e-4.
    ``chunk``: the backward walks the grid in voxel chunks (its workspace does not grow with the grid) - four chunks of the
    8^3 grid here, the parameter gradients adding up across them."""
    import tests.gpu_utils as gu
    if chunk:
        monkeypatch.setenv("HOLO_MLP_MEAN_BWD_CHUNK", chunk)
    g, (R, n_src, F, dim_out, n_hidden, n_harm), pick = _ref_mlp_mean_backward_fixture()
    cams_d, maps, sd = pick("cam."), pick("maps."), pick("param.")
    model = hda.HoloDiffusionModel(
        resol=R, feature_size=F, view_pooler_enabled=True, net_3d_enabled=False, diffusion_enabled=False,
        render_image_width=8, render_image_height=8, volume_extent=float(g["volume_extent"]),
        view_pooler_args=dict(feature_aggregator_class_type="MLPMeanFeatureAggregator",
                              feature_aggregator_MLPMeanFeatureAggregator_args=dict(
                                  n_hidden=n_hidden, dim_out=dim_out, n_harmonic_functions_ray=n_harm)))
    full = {"pooled_feature_mapper.weight": pick("mapper.")["weight"], "pooled_feature_mapper.bias": pick("mapper.")["bias"]}
    full.update({"view_pooler.feature_aggregator." + k: v for k, v in sd.items()})
    res = model.load_state_dict(full, strict=False)
    assert not res.unexpected_keys
    model.to(gu.DEV)
    cams = hda.PerspectiveCameras(R=cams_d["R"], T=cams_d["T"], focal_length=cams_d["focal"], principal_point=cams_d["pp"])
    dev = {k: v.to(gu.DEV) for k, v in maps.items()}
    out = model.pool_views_to_voxel_features(dev, cams.to(gu.DEV))
    assert (out.cpu() - torch.from_numpy(g["out"])).abs().max() < 1e-4
    got = model.pool_views_backward(dev, cams.to(gu.DEV), torch.from_numpy(g["cot"]).to(gu.DEV))
    rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12))  # noqa: E731
    assert rel(got["pooled_feature_mapper"]["weight"], pick("grad.mapper.")["weight"]) < 1e-3
    assert rel(got["pooled_feature_mapper"]["bias"], pick("grad.mapper.")["bias"]) < 1e-3
    ref_p = pick("grad.param.")
    assert set(got["feature_aggregator"]) == set(ref_p)
    for k, r in ref_p.items():
        assert rel(got["feature_aggregator"][k], r) < 1e-3, (k, rel(got["feature_aggregator"][k], r))
    for k, r in pick("grad.maps.").items():
        assert rel(got["image_features"][k], r) < 1e-3, (k, rel(got["image_features"][k], r))
    # the deterministic mode (holo_ctx_set_deterministic: fixed-point scatter, one binary point per chunk and map): bit-identical
    # from call to call, within fp32 summation noise of the default mode, parameter gradients untouched
    from holo_diffusion_amd import runtime
    with runtime.deterministic(True, gu.DEV):
        d1 = model.pool_views_backward(dev, cams.to(gu.DEV), torch.from_numpy(g["cot"]).to(gu.DEV))
        d1 = {grp: {k: v.clone() for k, v in d1[grp].items()} for grp in ("image_features", "feature_aggregator")}
        d2 = model.pool_views_backward(dev, cams.to(gu.DEV), torch.from_numpy(g["cot"]).to(gu.DEV))
    for k in d1["feature_aggregator"]:
        assert torch.equal(d1["feature_aggregator"][k], got["feature_aggregator"][k]), k
    for k, v in d1["image_features"].items():
        assert torch.equal(v, d2["image_features"][k]), k
        assert rel(v, got["image_features"][k].cpu()) < 2e-5, (k, rel(v, got["image_features"][k].cpu()))


@pytest.mark.gpu
def test_reconstruction_flyaround_from_a_dataset_sequence(tmp_path):
    """render_flyaround(dataset, sequence_name, model, sample_mode=False) - the reconstruction mode of
    visualize_reconstruction.py:60-162 / flyaround.py:153-171,219-253: source frames drawn with the reference's seeded
    permutation, pooled once (MLPMeanFeatureAggregator here), refined, rendered from the simple-360 cameras.  Equals the
    per-frame `model(camera=[target, sources...], image_features=...)` calls the reference's loop makes."""
    import tests.gpu_utils as gu
    from holo_diffusion_amd.generate import render_flyaround, select_source_views
    R, F, n_frames, n_src = 8, 16, 7, 3
    model, *_ = gu.make_model(R, F, 10, 12, dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)))
    m2 = hda.HoloDiffusionModel(resol=R, feature_size=F, render_image_width=12, render_image_height=10, view_pooler_enabled=True,
                                view_pooler_args=dict(feature_aggregator_class_type="MLPMeanFeatureAggregator"),
                                net_3d_SimpleUnet3D_args=dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2)))
    D = 16 + 1 + 3 + 21
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    sd.update({"view_pooler.feature_aggregator." + k: v for k, v in synth_state_dict(vo.mlp_mean_param_shapes(D), 31).items()})
    sd.update(synth_state_dict({"pooled_feature_mapper.weight": (F, 128), "pooled_feature_mapper.bias": (F,)}, 5))
    m2.load_state_dict(sd)
    m2.to(gu.DEV)
    feats, _ = _synthetic_views(n_frames, 300)
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, n_frames, -0.4, 9, (0.0, -1.0, 0.0), 3.0)

    class Frame:
        def __init__(self, i):
            self.camera = cams[i]
            self.image_features = {k: v[i] for k, v in feats.items()}
            self.sequence_name = "seq_a"

    class Dataset:
        def sequence_indices_in_order(self, name):
            assert name == "seq_a"
            return iter(range(100, 100 + n_frames))

        def __getitem__(self, idx):
            return Frame(idx - 100)

    out = render_flyaround(Dataset(), "seq_a", m2, str(tmp_path / "video"), n_flyaround_poses=3, trajectory_type="simple_360",
                           n_source_views=n_src, seed=11, sample_mode=False, device=gu.DEV,
                           output_video_frames_dir=str(tmp_path / "frames"))
    assert out["images_render"].shape == (3, 3, 10, 12) and torch.isfinite(out["images_render"]).all()
    assert os.path.isfile(os.path.join(str(tmp_path / "frames"), "seq_a_images_render", "frame_00002.ppm"))
    sel = select_source_views(n_frames, n_src, 11)
    assert len(set(sel)) == n_src and sel == select_source_views(n_frames, n_src, 11)
    test_cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 3, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2)
    src = cams[sel]
    for n in range(3):  # the reference's loop: batch = [target camera, sources...], one model() call per pose
        both = hda.PerspectiveCameras(R=torch.cat([test_cams.R[n:n + 1], src.R]), T=torch.cat([test_cams.T[n:n + 1], src.T]),
                                      focal_length=torch.cat([test_cams.focal_length[n:n + 1], src.focal_length]),
                                      principal_point=torch.cat([test_cams.principal_point[n:n + 1], src.principal_point]))
        preds = m2(camera=both.to(gu.DEV), image_features={k: v[sel].to(gu.DEV) for k, v in feats.items()})
        assert torch.equal(preds["images_render"][0], out["images_render"][n])
    with pytest.raises(ValueError):
        render_flyaround(None, "seq_a", m2, "", trajectory_type="simple_360", sample_mode=False, device=gu.DEV)
