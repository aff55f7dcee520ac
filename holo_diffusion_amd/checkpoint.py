"""Experiment-directory ingestion: ``expconfig.yaml`` + the last ``model_epoch_*.pth`` checkpoint.

Mirrors what the reference's sampling scripts do before they touch the hot path:

* ``holo_diffusion/utils/checkpoint_utils.py:16-76`` (``load_experiment``): read ``<exp_dir>/expconfig.yaml``,
  override the render size, force-resume the model factory;
* ``trainer/model_factory.py:73-133`` (``ImplicitronModelFactory.__call__``): pick ``resume_epoch`` or the last
  checkpoint of the directory, ``load_state_dict(strict=True)`` with a non-strict retry;
* PyTorch3D ``model_io`` naming: ``model_epoch_%08d.pth`` (optimizer state lives in ``*_opt.pth``, ignored here).

Only ``model_factory_ImplicitronModelFactory_args.model_HoloDiffusionModel_args`` is consumed.  Fields of components
outside the denoise-and-render path (image encoder, view pooling, metrics, losses, data source, optimizer, training
loop) are accepted and reported as ``ignored`` instead of being instantiated; checkpoint tensors that belong to them
are reported as ``unexpected``.  No OmegaConf / PyTorch3D needed (plain ``yaml``).
"""
from __future__ import annotations

import glob
import logging
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch
import yaml

from .registry import config_fields

logger = logging.getLogger(__name__)

MODEL_FACTORY_KEY = "model_factory_ImplicitronModelFactory_args"
MODEL_ARGS_KEY = "model_HoloDiffusionModel_args"
# state-dict prefixes that ARE the hot path: everything else in a reference checkpoint is encoder side
PATH_PREFIXES = ("net_3d.", "_implicit_functions.")
DEFAULT_VIEW_METRICS = "ViewMetrics"  # Implicitron's default `view_metrics_class_type`


@dataclass
class LoadReport:
    config_file: str = ""
    checkpoint_file: Optional[str] = None
    ignored_config_fields: List[str] = field(default_factory=list)
    missing_keys: List[str] = field(default_factory=list)
    unexpected_keys: List[str] = field(default_factory=list)
    strict: bool = False


def read_expconfig(exp_dir_or_file: str) -> Tuple[Dict[str, Any], str]:
    """``_get_config_from_experiment_directory`` (checkpoint_utils.py:16-19); also accepts the YAML file itself."""
    fn = exp_dir_or_file
    if os.path.isdir(fn):
        fn = os.path.join(fn, "expconfig.yaml")
    with open(fn) as f:
        cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict):
        raise ValueError(f"{fn}: not an experiment config")
    return cfg, fn


def _filter_fields(cls, args: Optional[Dict[str, Any]], where: str, ignored: List[str]) -> Dict[str, Any]:
    fields = config_fields(cls)
    out = {}
    for k, v in (args or {}).items():
        if k in fields:
            out[k] = v
        else:
            ignored.append(f"{where}.{k}")
    return out


def model_args_from_expconfig(cfg: Dict[str, Any], render_size: Optional[Tuple[int, int]] = None
                              ) -> Tuple[Dict[str, Any], List[str]]:
    """Constructor kwargs of :class:`HoloDiffusionModel` from a reference experiment config.

    ``render_size = (width, height)`` overrides ``render_image_width/height`` like ``load_experiment(render_size=…)``
    (checkpoint_utils.py:57-64).  Returns ``(kwargs, ignored_fields)``."""
    from .diffusion import ImplicitronGaussianDiffusion
    from .model import HoloDiffusionModel
    from .render import (AdaptiveRaySampler, HoloMultiPassEmissionAbsorptionRenderer, HoloVoxelGridImplicitFunction,
                         RenderMLP)
    from .unet import SimpleUnet3D

    factory = cfg.get(MODEL_FACTORY_KEY) or {}
    mtype = factory.get("model_class_type", "HoloDiffusionModel")
    if mtype != "HoloDiffusionModel":
        raise ValueError(f"model_class_type '{mtype}': only HoloDiffusionModel experiments are supported")
    margs = factory.get(MODEL_ARGS_KEY)
    if margs is None:
        raise ValueError(f"experiment config has no {MODEL_FACTORY_KEY}.{MODEL_ARGS_KEY}")
    ignored: List[str] = []
    margs = dict(margs)
    # view metrics are outside the denoise-and-render path (never instantiated).  unet_with_no_diffusion.yaml:183-185 names
    # `HoloDiffusionMetrics`, a class that exists nowhere in the released code: the default stands in, and the report says so
    vm = margs.pop("view_metrics_class_type", None)
    if vm is not None:
        ignored.append(f"{MODEL_ARGS_KEY}.view_metrics_class_type" + (
            "" if vm == DEFAULT_VIEW_METRICS else f" ('{vm}' is not a class of the released code -> {DEFAULT_VIEW_METRICS})"))
    kw = _filter_fields(HoloDiffusionModel, margs, MODEL_ARGS_KEY, ignored)
    nested = {
        "net_3d_SimpleUnet3D_args": SimpleUnet3D,
        "diffusion_args": ImplicitronGaussianDiffusion,
        "raysampler_AdaptiveRaySampler_args": AdaptiveRaySampler,
        "renderer_HoloMultiPassEmissionAbsorptionRenderer_args": HoloMultiPassEmissionAbsorptionRenderer,
        "implicit_function_HoloVoxelGridImplicitFunction_args": HoloVoxelGridImplicitFunction,
    }
    for key, cls in nested.items():
        if kw.get(key) is not None:
            kw[key] = _filter_fields(cls, kw[key], f"{MODEL_ARGS_KEY}.{key}", ignored)
    ifa = kw.get("implicit_function_HoloVoxelGridImplicitFunction_args")
    if ifa and ifa.get("render_mlp_args") is not None:
        ifa["render_mlp_args"] = _filter_fields(
            RenderMLP, ifa["render_mlp_args"],
            f"{MODEL_ARGS_KEY}.implicit_function_HoloVoxelGridImplicitFunction_args.render_mlp_args", ignored)
    # encoder side (view pooling): kept when the configured pooler is one the fused kernels implement (the released
    # YAMLs: AngleWeightedReductionFeatureAggregator [AVG, STD] or the learnt MLPMeanFeatureAggregator of hydrant.yaml /
    # old_base_config.yaml; bilinear, unmasked), dropped - and reported - otherwise, so that sampling from ANY
    # HoloDiffusion checkpoint keeps working
    if kw.get("view_pooler_enabled"):
        from .viewpool import (AngleWeightedReductionFeatureAggregator, MLPMeanFeatureAggregator, ViewPooler,
                               ViewSampler)
        vpa = _filter_fields(ViewPooler, kw.get("view_pooler_args") or {}, f"{MODEL_ARGS_KEY}.view_pooler_args", ignored)
        try:
            if vpa.get("view_sampler_args") is not None:
                vpa["view_sampler_args"] = _filter_fields(ViewSampler, vpa["view_sampler_args"],
                                                          f"{MODEL_ARGS_KEY}.view_pooler_args.view_sampler_args", ignored)
            akey = "feature_aggregator_AngleWeightedReductionFeatureAggregator_args"
            if vpa.get(akey) is not None:
                vpa[akey] = _filter_fields(AngleWeightedReductionFeatureAggregator, vpa[akey],
                                           f"{MODEL_ARGS_KEY}.view_pooler_args.{akey}", ignored)
            mkey = "feature_aggregator_MLPMeanFeatureAggregator_args"
            if vpa.get(mkey) is not None:
                vpa[mkey] = _filter_fields(MLPMeanFeatureAggregator, vpa[mkey], f"{MODEL_ARGS_KEY}.view_pooler_args.{mkey}",
                                           ignored)
            ViewPooler(**vpa)  # validates the configuration
            kw["view_pooler_args"] = vpa
        except (NotImplementedError, ValueError, TypeError, KeyError) as e:
            ignored.append(f"{MODEL_ARGS_KEY}.view_pooler_enabled (view pooling disabled: {e})")
            kw["view_pooler_enabled"] = False
            kw.pop("view_pooler_args", None)
    else:
        kw.pop("view_pooler_args", None)
    if render_size is not None:
        kw["render_image_width"], kw["render_image_height"] = int(render_size[0]), int(render_size[1])
    return kw, ignored


def find_last_checkpoint(exp_dir: str) -> Optional[str]:
    """PyTorch3D ``model_io.find_last_checkpoint``: lexicographically last ``model_epoch_<8 digits>.pth``."""
    pat = os.path.join(glob.escape(exp_dir), "model_epoch_" + "[0-9]" * 8 + ".pth")
    fls = sorted(glob.glob(pat))
    return fls[-1] if fls else None


def get_checkpoint(exp_dir: str, epoch: int) -> str:
    return os.path.join(exp_dir, "model_epoch_%08d.pth" % epoch)


def load_model_state(model: torch.nn.Module, state: Dict[str, torch.Tensor], report: Optional[LoadReport] = None
                     ) -> LoadReport:
    """``load_state_dict(strict=True)`` with the factory's non-strict retry (model_factory.py:118-126).  Tensors
    of the encoder side of a reference checkpoint end up in ``report.unexpected_keys``; a hot-path parameter that
    the checkpoint lacks is an error (the model would silently keep its initialisation otherwise)."""
    report = report or LoadReport()
    try:
        model.load_state_dict(state, strict=True)
        report.strict = True
        return report
    except RuntimeError as e:
        logger.info("Cannot load state dict in strict mode (%s) -> trying non-strict", str(e).split("\n")[0])
    res = model.load_state_dict(state, strict=False)
    report.missing_keys = list(res.missing_keys)
    report.unexpected_keys = list(res.unexpected_keys)
    bad = [k for k in report.missing_keys if k.startswith(PATH_PREFIXES)]
    if bad:
        raise KeyError(f"checkpoint lacks {len(bad)} parameters of the denoise/render path, e.g. {bad[:3]}")
    return report


def load_experiment(exp_dir: str, render_size: Optional[Tuple[int, int]] = None, device: Any = None,
                    resume_epoch: int = -1, force_resume: bool = True):
    """Build :class:`HoloDiffusionModel` from ``<exp_dir>/expconfig.yaml`` and load its last checkpoint.

    Returns ``(model, report)``.  Same decision table as ``ImplicitronModelFactory.__call__``
    (trainer/model_factory.py:96-133): a found checkpoint is loaded when ``force_resume`` (the value
    ``load_experiment`` of the reference sets, checkpoint_utils.py:60) or the config's ``resume`` flag (default True)
    is set, otherwise the model keeps its initialisation ("Not resuming -> starting from scratch"); a missing
    checkpoint is a ``FileNotFoundError`` only under ``force_resume``."""
    from .model import HoloDiffusionModel
    cfg, fn = read_expconfig(exp_dir)
    kw, ignored = model_args_from_expconfig(cfg, render_size)
    report = LoadReport(config_file=fn, ignored_config_fields=ignored)
    model = HoloDiffusionModel(**kw)
    factory = cfg.get(MODEL_FACTORY_KEY) or {}
    if resume_epoch <= 0:
        resume_epoch = int(factory.get("resume_epoch", -1) or -1)
    if resume_epoch > 0:
        path = get_checkpoint(exp_dir, resume_epoch)
        if not os.path.isfile(path):
            raise ValueError(f"Cannot find model from epoch {resume_epoch}.")
    else:
        path = find_last_checkpoint(exp_dir)
    resume = bool(factory.get("resume", True))
    if path is not None:
        if force_resume or resume:
            state = torch.load(path, map_location="cpu", weights_only=True)
            report.checkpoint_file = path
            load_model_state(model, state, report)
        else:
            logger.info("Found %s but not resuming -> starting from scratch.", path)
    elif force_resume:
        raise FileNotFoundError(f"Cannot find a checkpoint in {exp_dir}!")
    if device is not None:
        model.to(device)
    return model, report
