"""The plugin classes join the REAL Implicitron class tree and registry when PyTorch3D is importable
(holo_diffusion_amd/registry.py).  PyTorch3D is not installable here, so a minimal fake of its config module
(tests/support/fake_pytorch3d.py) is injected into sys.modules of a fresh interpreter before the package is imported."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, {repo!r})
from tests.support import fake_pytorch3d
p3d_registry = fake_pytorch3d.install()
import torch
import holo_diffusion_amd as hda
import pytorch3d.implicitron.tools.config as p3dc
from pytorch3d.implicitron.models.base_model import ImplicitronModelBase
from pytorch3d.implicitron.models.renderer.base import BaseRenderer
from pytorch3d.implicitron.models.implicit_function.base import ImplicitFunctionBase

assert hda.HAVE_PYTORCH3D
assert hda.ReplaceableBase is p3dc.ReplaceableBase and hda.Configurable is p3dc.Configurable
# real members of the Implicitron class tree ...
assert issubclass(hda.HoloDiffusionModel, ImplicitronModelBase)
assert issubclass(hda.HoloMultiPassEmissionAbsorptionRenderer, BaseRenderer)
assert issubclass(hda.HoloVoxelGridImplicitFunction, ImplicitFunctionBase)
assert issubclass(hda.SimpleUnet3D, p3dc.ReplaceableBase) and issubclass(hda.ImplicitronGaussianDiffusion, p3dc.Configurable)
# ... that Implicitron's own registry resolves (what ImplicitronModelFactory / GenericModel do with *_class_type)
assert p3d_registry.get(ImplicitronModelBase, "HoloDiffusionModel") is hda.HoloDiffusionModel
assert p3d_registry.get(BaseRenderer, "HoloMultiPassEmissionAbsorptionRenderer") is hda.HoloMultiPassEmissionAbsorptionRenderer
assert p3d_registry.get(ImplicitFunctionBase, "HoloVoxelGridImplicitFunction") is hda.HoloVoxelGridImplicitFunction
assert p3d_registry.get(hda.Unet3DBase, "SimpleUnet3D") is hda.SimpleUnet3D
for c in (hda.HoloDiffusionModel, hda.SimpleUnet3D, hda.HoloVoxelGridImplicitFunction, hda.HoloMultiPassEmissionAbsorptionRenderer):
    assert hda.pytorch3d_registered(c) is True, c
# the package registry agrees
assert hda.registry.get(ImplicitronModelBase, "HoloDiffusionModel") is hda.HoloDiffusionModel

# a config-only class that shares its NAME with a PyTorch3D class stays out of PyTorch3D's registry (which overwrites
# silently): Implicitron's own AngleWeightedReductionFeatureAggregator must survive the import of this package ...
from pytorch3d.implicitron.models.view_pooler.feature_aggregator import FeatureAggregatorBase
own = p3d_registry.get(FeatureAggregatorBase, "AngleWeightedReductionFeatureAggregator")
assert getattr(own, "IS_PYTORCH3D_OWN", False) and hasattr(own, "forward"), own
# ... while the package resolves the name to its own parameter holder (a member of PyTorch3D's class tree)
from holo_diffusion_amd.viewpool import AngleWeightedReductionFeatureAggregator as mine, ViewPooler
assert hda.registry.get(FeatureAggregatorBase, "AngleWeightedReductionFeatureAggregator") is mine and mine is not own
assert issubclass(mine, FeatureAggregatorBase)
assert type(ViewPooler().feature_aggregator) is mine

# instantiation goes through PyTorch3D's dataclass processing (ReplaceableBase.__new__) and still builds the plugin
model = hda.HoloDiffusionModel(resol=8, feature_size=16, render_image_width=12, render_image_height=10,
                               net_3d_SimpleUnet3D_args=dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(2,)),
                               diffusion_args=dict(num_steps=250))
assert getattr(hda.HoloDiffusionModel, "_processed_by_fake_pytorch3d", False)
assert getattr(hda.SimpleUnet3D, "_processed_by_fake_pytorch3d", False)
assert model.resol == 8 and model.net_3d.image_size == 8 and model.net_3d.in_channels == 16 and model.diffusion.num_steps == 250
keys = list(model.state_dict())
assert any(k.startswith("net_3d._net.input_blocks.0.0.") for k in keys)
assert any(k.startswith("_implicit_functions.0._fn.render_mlp._density_net.mlp.0.0.") for k in keys)
assert hda.get_default_args(hda.SimpleUnet3D)["channel_mult"] == [1, 2, 4, 8]   # defaults readable after processing
try:
    hda.HoloDiffusionModel(no_such_field=1)
    raise SystemExit("unknown config field accepted")
except TypeError:
    pass

# a registration PyTorch3D rejects is an ERROR (round 1 swallowed it)
class NotReplaceable(torch.nn.Module):
    pass
try:
    hda.registry.register(NotReplaceable)
    raise SystemExit("registering a non-replaceable class did not raise")
except ValueError:
    pass
print("OK")
"""


def test_plugins_register_in_the_pytorch3d_registry():
    res = subprocess.run([sys.executable, "-c", SCRIPT.format(repo=REPO)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout[-2000:] + res.stderr[-4000:]


def test_without_pytorch3d_the_local_registry_resolves_the_plugins():
    import holo_diffusion_amd as hda
    from holo_diffusion_amd.model import ImplicitronModelBase
    from holo_diffusion_amd.render import BaseRenderer, ImplicitFunctionBase
    if hda.HAVE_PYTORCH3D:
        return
    assert hda.pytorch3d_registered(hda.HoloDiffusionModel) is None
    assert hda.registry.get(ImplicitronModelBase, "HoloDiffusionModel") is hda.HoloDiffusionModel
    assert hda.registry.get(BaseRenderer, "HoloMultiPassEmissionAbsorptionRenderer") is hda.HoloMultiPassEmissionAbsorptionRenderer
    assert hda.registry.get(ImplicitFunctionBase, "HoloVoxelGridImplicitFunction") is hda.HoloVoxelGridImplicitFunction
    assert hda.registry.get(hda.Unet3DBase, "SimpleUnet3D") is hda.SimpleUnet3D
    try:
        hda.registry.get(hda.Unet3DBase, "NoSuchNet")
        raise AssertionError("unknown plugin name resolved")
    except ValueError:
        pass
