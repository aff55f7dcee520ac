"""holo_diffusion_amd — MI355X-native denoise-and-render hot path of HoloDiffusion.

Plugin surface (same names as the reference's Implicitron plugins):
``SimpleUnet3D``, ``ImplicitronGaussianDiffusion``, ``HoloVoxelGridImplicitFunction``,
``HoloMultiPassEmissionAbsorptionRenderer``, ``HoloDiffusionModel``; all arithmetic runs in
``libholo_mi355x.so`` (C ABI in ``include/holo_abi.h``).  There is no CPU fallback.
"""
from .registry import (registry, get_default_args, config_fields, Configurable, ReplaceableBase,  # noqa: F401
                       HAVE_PYTORCH3D, pytorch3d_registered)
from .unet import SimpleUnet3D, Unet3DBase  # noqa: F401
from .diffusion import ImplicitronGaussianDiffusion, ModelMeanType, ModelVarType  # noqa: F401
from .render import (  # noqa: F401
    EvaluationMode, HoloMultiPassEmissionAbsorptionRenderer, HoloVoxelGridImplicitFunction, ImplicitFunctionWrapper,
    RenderMLP, RendererOutput, ImplicitronRayBundle, AdaptiveRaySampler)
from .model import HoloDiffusionModel  # noqa: F401
from .cameras import PerspectiveCameras, look_at_view_transform, get_simple_360_camera_trajectory  # noqa: F401
from .viewpool import ViewPooler, AngleWeightedReductionFeatureAggregator  # noqa: F401,E402
from . import checkpoint, flyaround_output, generate, model, render, runtime, viewpool  # noqa: F401,E402
from .checkpoint import load_experiment  # noqa: F401,E402

__all__ = ["registry", "SimpleUnet3D", "Unet3DBase", "ImplicitronGaussianDiffusion", "HoloVoxelGridImplicitFunction",
           "HoloMultiPassEmissionAbsorptionRenderer", "HoloDiffusionModel", "RenderMLP", "EvaluationMode",
           "PerspectiveCameras", "get_simple_360_camera_trajectory"]
