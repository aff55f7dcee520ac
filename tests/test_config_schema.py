"""CPU: the config ingestion of SURVEY.md 8(f1) against the SCHEMA of the reference's own released experiment configs
(configs/{apple,hydrant,teddybear,donut,unet_with_no_diffusion}.yaml; apple.yaml:68-253 is the model block).
``tests/golden/ref_config_keys.json`` is recorded from those files by ``oracle/make_golden_config.py`` (development
container only): every leaf under ``model_HoloDiffusionModel_args`` with its value, the model factory's fields, all
``*_class_type`` values, the names of everything else.  For each of the five configs: every key of the model block is
either consumed by a plugin class or reported as ignored; the ``*_class_type`` names resolve; the one class the released
code does not contain (``HoloDiffusionMetrics``) maps to the default; the model constructs
(holo_diffusion/utils/checkpoint_utils.py:23-76, trainer/model_factory.py:96-133)."""
import json
import os

import pytest

import holo_diffusion_amd as hda
from holo_diffusion_amd import checkpoint as ck

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("apple", "hydrant", "teddybear", "donut", "unet_with_no_diffusion")


@pytest.fixture(scope="module")
def schema():
    with open(os.path.join(REPO, "tests", "golden", "ref_config_keys.json")) as f:
        return json.load(f)


def _nest(leaves):
    root = {}
    for path, value in leaves:
        if isinstance(value, dict) and "__list_of_names__" in value:
            value = [f"name{i}" for i in range(value["__list_of_names__"])]
        node = root
        parts = path.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = value
    return root


def _expconfig(rec):
    cfg = {k: ({kk: None for kk in v} if v else None) for k, v in rec["top_level"].items()}
    cfg[ck.MODEL_FACTORY_KEY] = dict(rec["model_factory"])
    cfg[ck.MODEL_FACTORY_KEY][ck.MODEL_ARGS_KEY] = _nest(rec["model_args"])
    return cfg


def _flat(node, prefix=""):
    out = {}
    for k, v in node.items():
        path = f"{prefix}.{k}" if prefix else k
        if isinstance(v, dict) and v:
            out.update(_flat(v, path))
        else:
            out[path] = v
    return out


def test_fixture_covers_the_five_released_configs(schema):
    assert sorted(schema) == sorted(NAMES)
    for name in NAMES:
        rec = schema[name]
        assert rec["model_factory"]["model_class_type"] == "HoloDiffusionModel"
        assert len(rec["model_args"]) >= 60
        assert "data_source_ImplicitronDataSource_args" in rec["top_level"]  # the parts outside the path are named, not kept


@pytest.mark.parametrize("name", NAMES)
def test_every_key_of_the_model_block_is_consumed_or_reported(schema, name):
    rec = schema[name]
    kw, ignored = ck.model_args_from_expconfig(_expconfig(rec))
    consumed = _flat(kw)
    ign = [s.split(" ")[0] for s in ignored]  # (an entry may carry an explanation behind the key path)
    pre = ck.MODEL_ARGS_KEY + "."
    unexplained = []
    for path, value in rec["model_args"]:
        if path in consumed or (isinstance(value, dict) and not value and path in consumed):
            if not (isinstance(value, dict) and "__list_of_names__" in value):
                got = consumed[path]
                assert (list(got) if isinstance(got, tuple) else got) == value, (path, got, value)
            continue
        full = pre + path
        if any(full == i or full.startswith(i + ".") for i in ign):
            continue
        unexplained.append(path)
    assert not unexplained, unexplained
    # nothing the hot path needs went into the ignored list
    hot = ("resol", "feature_size", "net_3d_SimpleUnet3D_args.", "diffusion_args.", "raysampler_AdaptiveRaySampler_args.n_pts",
           "renderer_HoloMultiPassEmissionAbsorptionRenderer_args.n_pts", "implicit_function_HoloVoxelGridImplicitFunction_args.render_mlp_args.")
    assert not [i for i in ign if i[len(pre):].startswith(hot)], ign


@pytest.mark.parametrize("name", NAMES)
def test_class_types_resolve_and_the_model_constructs(schema, name):
    rec = schema[name]
    types = {k.split(ck.MODEL_ARGS_KEY + ".")[-1]: v for k, v in rec["class_types"].items() if ck.MODEL_ARGS_KEY in k}
    assert types["net_3d_class_type"] == "SimpleUnet3D" and types["raysampler_class_type"] == "AdaptiveRaySampler"
    assert types["renderer_class_type"] == "HoloMultiPassEmissionAbsorptionRenderer"
    assert types["implicit_function_class_type"] == "HoloVoxelGridImplicitFunction"
    kw, ignored = ck.model_args_from_expconfig(_expconfig(rec))
    vm = [s for s in ignored if "view_metrics_class_type" in s]
    assert len(vm) == 1
    if types["view_metrics_class_type"] == "HoloDiffusionMetrics":  # unet_with_no_diffusion.yaml:183-185
        assert "-> ViewMetrics" in vm[0]
    else:
        assert types["view_metrics_class_type"] == "ViewMetrics" and "->" not in vm[0]
    margs = dict(rec["model_args"])
    # the encoder side stays on where the YAML enables it: both released aggregators are implemented
    assert kw.get("view_pooler_enabled", False) == bool(margs.get("view_pooler_enabled", False))
    model = hda.HoloDiffusionModel(**kw)
    if kw.get("view_pooler_enabled"):  # (PyTorch3D ViewPooler's default aggregator where the YAML names none)
        want = types.get("view_pooler_args.feature_aggregator_class_type", "AngleWeightedReductionFeatureAggregator")
        assert type(model.view_pooler.feature_aggregator).__name__ == want
    assert model.resol == margs["resol"] and model.feature_size == margs["feature_size"]
    assert model.net_3d_enabled == margs["net_3d_enabled"] and model.diffusion_enabled == margs["diffusion_enabled"]
    if model.net_3d_enabled:
        assert model.net_3d.model_channels == margs["net_3d_SimpleUnet3D_args.model_channels"]
        assert tuple(model.net_3d.channel_mult) == tuple(margs["net_3d_SimpleUnet3D_args.channel_mult"])
        # HoloDiffusionModel overrides the YAML's in/out channels and image size (holo_diffusion_model.py:118-130)
        assert model.net_3d.in_channels == model.feature_size and model.net_3d.image_size == model.resol
    if model.diffusion_enabled:
        assert model.diffusion.num_steps == margs["diffusion_args.num_steps"]
    assert model.renderer.n_pts_per_ray_fine_evaluation == \
        margs["renderer_HoloMultiPassEmissionAbsorptionRenderer_args.n_pts_per_ray_fine_evaluation"]
    assert model.render_image_width == margs["render_image_width"]
    # state-dict names of the path as a reference checkpoint spells them (trainer/model_factory.py:115-126)
    keys = list(model.state_dict())
    assert any(k.startswith("_implicit_functions.0._fn.render_mlp._density_net.mlp.0.0.") for k in keys)
    if model.net_3d_enabled:
        assert any(k.startswith("net_3d._net.input_blocks.0.0.") for k in keys)
