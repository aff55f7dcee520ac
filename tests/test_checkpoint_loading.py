"""Experiment-directory ingestion: expconfig.yaml schema + last-checkpoint state-dict loading.

Reference behaviour followed: holo_diffusion/utils/checkpoint_utils.py:16-76, trainer/model_factory.py:73-133,
config schema of configs/apple.yaml:68-253 (keys only; the YAML is generated here, values are small)."""
import os

import pytest
import torch
import yaml

import holo_diffusion_amd as hda
from holo_diffusion_amd import checkpoint as ck
from holo_diffusion_amd.weights import synth_state_dict


def _expconfig(resol=8, feat=16, mc=32):
    margs = {
        # ---- fields of components outside the hot path (must be accepted and reported, never instantiated)
        "log_vars": ["loss_rgb_psnr", "objective", "epoch"],
        "mask_images": True, "mask_depths": True, "mask_threshold": 0.5, "output_rasterized_mc": True,
        "render_features_dimensions": 3, "tqdm_trigger_threshold": 100, "global_encoder_class_type": None,
        "image_feature_extractor_class_type": "ResNetFeatureExtractor", "view_pooler_enabled": True,
        "view_metrics_class_type": "ViewMetrics", "regularization_metrics_class_type": "RegularizationMetrics",
        "loss_weights": {"loss_rgb_mse": 1.0, "loss_mask_bce": 0.0},
        "image_feature_extractor_ResNetFeatureExtractor_args": {"name": "resnet34", "stages": [1, 2, 3, 4]},
        "view_pooler_args": {"feature_aggregator_class_type": "AngleWeightedReductionFeatureAggregator"},
        # ---- hot-path fields
        "render_image_width": 256, "render_image_height": 256, "bg_color": [1.0, 1.0, 1.0], "num_passes": 2,
        "chunk_size_grid": 163840, "n_train_target_views": 10,
        "sampling_mode_training": "mask_sample", "sampling_mode_evaluation": "full_grid",
        "raysampler_class_type": "AdaptiveRaySampler", "renderer_class_type": "HoloMultiPassEmissionAbsorptionRenderer",
        "implicit_function_class_type": "HoloVoxelGridImplicitFunction",
        "raysampler_AdaptiveRaySampler_args": {"n_pts_per_ray_evaluation": 64, "scene_extent": 4.0,
                                               "n_rays_per_image_sampled_from_mask": 1024, "cast_ray_bundle_as_cone": False},
        "renderer_HoloMultiPassEmissionAbsorptionRenderer_args": {
            "n_pts_per_ray_fine_evaluation": 16, "append_coarse_samples_to_fine": True,
            "raymarcher_EmissionAbsorptionRaymarcher_args": {"background_opacity": 1e10, "blend_output": False}},
        "implicit_function_HoloVoxelGridImplicitFunction_args": {
            "resol": 32, "volume_extent": 8.0, "n_hidden": 128, "feature_dim": 64, "init_density_bias": 1e-4,
            "render_normals": True,
            "render_mlp_args": {"input_dims": 128, "output_feature_dims": 3, "dir_emb_dims": 4, "dnet_num_layers": 4,
                                "dnet_hidden_dim": 256, "dnet_input_skips": [2], "rnet_num_layers": 1,
                                "rnet_hidden_dim": 128, "rnet_input_skips": [], "activation_fn": "LEAKYRELU"}},
        "resol": resol, "volume_extent": 8.0, "feature_size": feat, "net_3d_enabled": True,
        "net_3d_class_type": "SimpleUnet3D", "diffusion_enabled": True, "enable_bootstrap": True, "bootstrap_prob": 0.5,
        "net_3d_SimpleUnet3D_args": {"image_size": 64, "in_channels": 128, "out_channels": 128, "model_channels": mc,
                                     "num_res_blocks": 2, "channel_mult": [1, 2], "attention_resolutions": [2],
                                     "num_heads": 2, "dropout": 0.0, "homogeneous_resample": True},
        "diffusion_args": {"beta_schedule_type": "linear", "num_steps": 250, "beta_start_unscaled": 1e-4,
                           "beta_end_unscaled": 0.02, "model_mean_type": "START_X", "model_var_type": "FIXED_SMALL",
                           "schedule_sampler_type": "uniform"},
    }
    return {
        "exp_dir": "/somewhere/else", "seed": 3,
        "data_source_ImplicitronDataSource_args": {"dataset_map_provider_class_type": "JsonIndexDatasetMapProviderV2"},
        "model_factory_ImplicitronModelFactory_args": {"resume": True, "model_class_type": "HoloDiffusionModel",
                                                       "resume_epoch": -1, "force_resume": False,
                                                       "model_HoloDiffusionModel_args": margs},
        "optimizer_factory_ImplicitronOptimizerFactory_args": {"breed": "Adam", "lr": 4e-5},
        "training_loop_ImplicitronTrainingLoop_args": {"max_epochs": 1000},
    }


def _reference_like_state(model, seed):
    """A checkpoint as the reference trainer would have written it: hot-path tensors + encoder-side tensors."""
    # (pooled_feature_mapper is a LazyLinear: no shape until a checkpoint or the first pooled batch gives it one)
    lazy = torch.nn.parameter.UninitializedParameter
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not isinstance(v, lazy)}
    if getattr(model, "view_pooler_enabled", False):  # 2 x (64 ResNet + 1 mask + 3 image) aggregated features
        shapes["pooled_feature_mapper.weight"] = (model.feature_size, 136)
        shapes["pooled_feature_mapper.bias"] = (model.feature_size,)
    sd = synth_state_dict(shapes, seed)
    for k in list(sd):  # the passes share ONE implicit function (holo_diffusion_model.py:165-169): identical tensors
        if k.startswith("_implicit_functions.") and not k.startswith("_implicit_functions.0."):
            sd[k] = sd["_implicit_functions.0." + k.split(".", 2)[2]].clone()
    sd["image_feature_extractor.stem.0.weight"] = torch.zeros(4, 3, 3, 3)
    sd["view_pooler.feature_aggregator._dummy"] = torch.zeros(1)
    return sd


@pytest.fixture()
def exp_dir(tmp_path):
    d = tmp_path / "exp"
    d.mkdir()
    with open(d / "expconfig.yaml", "w") as f:
        yaml.safe_dump(_expconfig(), f)
    return str(d)


def test_model_args_from_expconfig_filters_out_of_scope_fields(exp_dir):
    cfg, fn = ck.read_expconfig(exp_dir)
    assert fn.endswith("expconfig.yaml")
    kw, ignored = ck.model_args_from_expconfig(cfg, render_size=(96, 64))
    assert kw["render_image_width"] == 96 and kw["render_image_height"] == 64  # load_experiment(render_size=...)
    assert kw["resol"] == 8 and kw["feature_size"] == 16
    assert "log_vars" not in kw and "image_feature_extractor_class_type" not in kw
    # the encoder side is kept: the configured pooler is the one the fused view-pooling kernel implements
    assert kw["view_pooler_enabled"] is True
    assert kw["view_pooler_args"]["feature_aggregator_class_type"] == "AngleWeightedReductionFeatureAggregator"
    assert any(s.endswith(".log_vars") for s in ignored)
    assert any(s.endswith("raysampler_AdaptiveRaySampler_args.cast_ray_bundle_as_cone") for s in ignored)
    assert kw["implicit_function_HoloVoxelGridImplicitFunction_args"]["render_mlp_args"]["activation_fn"] == "LEAKYRELU"


def test_load_experiment_picks_last_checkpoint_and_binds_every_path_parameter(exp_dir):
    cfg, _ = ck.read_expconfig(exp_dir)
    kw, _ = ck.model_args_from_expconfig(cfg)
    proto = hda.HoloDiffusionModel(**kw)
    old, new = _reference_like_state(proto, 1), _reference_like_state(proto, 2)
    torch.save(old, os.path.join(exp_dir, "model_epoch_00000003.pth"))
    torch.save(new, os.path.join(exp_dir, "model_epoch_00000012.pth"))
    torch.save({"not": "a model"}, os.path.join(exp_dir, "model_epoch_00000012_opt.pth"))  # optimizer file: ignored
    assert ck.find_last_checkpoint(exp_dir).endswith("model_epoch_00000012.pth")
    model, rep = ck.load_experiment(exp_dir, render_size=(32, 24))
    assert rep.checkpoint_file.endswith("model_epoch_00000012.pth") and not rep.strict
    assert sorted(rep.unexpected_keys) == ["image_feature_extractor.stem.0.weight", "view_pooler.feature_aggregator._dummy"]
    assert rep.missing_keys == []
    # the encoder-side mapper of the checkpoint lands in the (lazy) pooled_feature_mapper
    assert tuple(model.pooled_feature_mapper.weight.shape) == (model.feature_size, 136)
    assert torch.equal(model.pooled_feature_mapper.weight.detach().cpu(), new["pooled_feature_mapper.weight"])
    got = model.state_dict()
    for k, v in new.items():
        if k.startswith(ck.PATH_PREFIXES):
            assert torch.equal(got[k], v), k
    # config reached the plugins: overrides of HoloDiffusionModel (image_size/in_channels from resol/feature_size,
    # holo_diffusion_model.py:118-130), enum fields by name, the 250-step schedule
    assert model.net_3d.image_size == 8 and model.net_3d.in_channels == 16 and model.net_3d.model_channels == 32
    assert model.diffusion.num_steps == 250 and model.diffusion.model_mean_type == hda.ModelMeanType.START_X
    assert model.render_image_width == 32 and model.render_image_height == 24
    assert model.renderer.n_pts_per_ray_fine_evaluation == 16
    # an explicit epoch
    model3, rep3 = ck.load_experiment(exp_dir, resume_epoch=3)
    assert rep3.checkpoint_file.endswith("model_epoch_00000003.pth")
    k0 = next(k for k in old if k.startswith("net_3d."))
    assert torch.equal(model3.state_dict()[k0], old[k0])


def test_missing_checkpoint_and_incomplete_checkpoint_are_errors(exp_dir):
    with pytest.raises(FileNotFoundError):
        ck.load_experiment(exp_dir)  # force_resume like the reference's load_experiment
    model, rep = ck.load_experiment(exp_dir, force_resume=False)
    assert rep.checkpoint_file is None
    sd = _reference_like_state(model, 5)
    k0 = next(k for k in sd if k.startswith("net_3d."))
    del sd[k0]
    torch.save(sd, os.path.join(exp_dir, "model_epoch_00000001.pth"))
    with pytest.raises(KeyError):
        ck.load_experiment(exp_dir)
    with pytest.raises(ValueError):
        ck.load_experiment(exp_dir, resume_epoch=7)


def test_wrong_model_class_is_rejected(exp_dir):
    cfg, _ = ck.read_expconfig(exp_dir)
    cfg["model_factory_ImplicitronModelFactory_args"]["model_class_type"] = "GenericModel"
    with pytest.raises(ValueError):
        ck.model_args_from_expconfig(cfg)


def test_resume_flags_follow_the_model_factory(exp_dir):
    """trainer/model_factory.py:96-133: a found checkpoint is loaded iff force_resume or the config's `resume`."""
    model, _ = ck.load_experiment(exp_dir, force_resume=False)
    sd = _reference_like_state(model, 11)
    torch.save(sd, os.path.join(exp_dir, "model_epoch_00000002.pth"))
    k0 = next(k for k in sd if k.startswith("net_3d."))
    # resume: true in the config -> loaded even without force_resume
    m1, r1 = ck.load_experiment(exp_dir, force_resume=False)
    assert r1.checkpoint_file is not None and torch.equal(m1.state_dict()[k0], sd[k0])
    # resume: false and no force_resume -> "Not resuming -> starting from scratch"
    cfg = _expconfig()
    cfg["model_factory_ImplicitronModelFactory_args"]["resume"] = False
    with open(os.path.join(exp_dir, "expconfig.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    m2, r2 = ck.load_experiment(exp_dir, force_resume=False)
    assert r2.checkpoint_file is None and not torch.equal(m2.state_dict()[k0], sd[k0])
    # force_resume (what the reference's load_experiment sets) overrides resume: false
    m3, r3 = ck.load_experiment(exp_dir)
    assert r3.checkpoint_file is not None and torch.equal(m3.state_dict()[k0], sd[k0])
