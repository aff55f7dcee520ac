"""TEST-ONLY stand-in for the parts of PyTorch3D's Implicitron config system the plugin registration touches
(pytorch3d 0.7.4, ``pytorch3d/implicitron/tools/config.py``; PyTorch3D itself is not installable here).

Mimicked semantics:
  * ``ReplaceableBase`` / ``Configurable``: marker classes whose ``__new__`` dataclass-processes the concrete class on
    first instantiation (``expand_args_fields`` -> ``dataclasses.dataclass(eq=False)``)
  * ``registry.register(cls)``: files ``cls`` under its most-base ancestor that derives from ``ReplaceableBase``
    (``ValueError`` when there is none); ``registry.get(base, name)``: ``base`` must derive DIRECTLY from
    ``ReplaceableBase`` (or have such an ancestor), unknown names raise ``ValueError``
  * the plugin base classes ``ImplicitronModelBase``, ``BaseRenderer``, ``ImplicitFunctionBase`` in their modules
``install()`` puts the fake modules into ``sys.modules``; it must run BEFORE ``holo_diffusion_amd`` is imported.
"""
import dataclasses
import sys
import types
from collections import defaultdict

import torch


def _is_actually_dataclass(cls) -> bool:
    return "__dataclass_fields__" in cls.__dict__


def expand_args_fields(cls):
    if _is_actually_dataclass(cls):
        return cls
    dataclasses.dataclass(eq=False)(cls)
    cls._processed_by_fake_pytorch3d = True
    return cls


class ReplaceableBase:
    def __new__(cls, *args, **kwargs):
        obj = super().__new__(cls)
        if cls is not ReplaceableBase and not _is_actually_dataclass(cls):
            expand_args_fields(cls)
        return obj


class Configurable:
    def __new__(cls, *args, **kwargs):
        obj = super().__new__(cls)
        if cls is not Configurable and not _is_actually_dataclass(cls):
            expand_args_fields(cls)
        return obj


class _Registry:
    def __init__(self):
        self._mapping = defaultdict(dict)

    @staticmethod
    def _is_base_class(some_class) -> bool:
        return ReplaceableBase in some_class.__bases__

    @staticmethod
    def _base_class_from_class(some_class):
        for base in some_class.mro()[-3::-1]:
            if base is not ReplaceableBase and issubclass(base, ReplaceableBase):
                return base
        return None

    def register(self, some_class):
        name = some_class.__name__
        base_class = self._base_class_from_class(some_class)
        if base_class is None:
            raise ValueError(f"Cannot register {some_class}. Cannot tell what it is.")
        self._mapping[base_class][name] = some_class
        return some_class

    def get(self, base_class_wanted, name: str):
        if self._is_base_class(base_class_wanted):
            base_class = base_class_wanted
        else:
            base_class = self._base_class_from_class(base_class_wanted)
            if base_class is None:
                raise ValueError(f"Cannot look up {base_class_wanted}. Cannot tell what it is.")
        result = self._mapping[base_class].get(name)
        if result is None:
            raise ValueError(f"{name} has not been registered.")
        if not issubclass(result, base_class_wanted):
            raise ValueError(f"{name} resolves to {result} which does not subclass {base_class_wanted}")
        return result


def install():
    registry = _Registry()
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m

    mod("pytorch3d")
    mod("pytorch3d.implicitron")
    mod("pytorch3d.implicitron.tools")
    mod("pytorch3d.implicitron.tools.config", registry=registry, ReplaceableBase=ReplaceableBase,
        Configurable=Configurable, expand_args_fields=expand_args_fields)
    mod("pytorch3d.implicitron.models")

    class ImplicitronModelBase(ReplaceableBase, torch.nn.Module):
        pass

    class BaseRenderer(ReplaceableBase):
        pass

    class ImplicitFunctionBase(ReplaceableBase):
        @staticmethod
        def allows_multiple_passes() -> bool:
            return False

    mod("pytorch3d.implicitron.models.base_model", ImplicitronModelBase=ImplicitronModelBase)
    mod("pytorch3d.implicitron.models.renderer")
    mod("pytorch3d.implicitron.models.renderer.base", BaseRenderer=BaseRenderer)
    mod("pytorch3d.implicitron.models.implicit_function")
    mod("pytorch3d.implicitron.models.implicit_function.base", ImplicitFunctionBase=ImplicitFunctionBase)
    # view pooler side: PyTorch3D ships its OWN AngleWeightedReductionFeatureAggregator (an nn.Module with a forward),
    # registered at import time; other Implicitron models in the same process depend on it
    class FeatureAggregatorBase(ReplaceableBase):
        pass

    class AngleWeightedReductionFeatureAggregator(torch.nn.Module, FeatureAggregatorBase):
        IS_PYTORCH3D_OWN = True

        def forward(self, *a, **k):
            return "pytorch3d's own aggregator"

    registry.register(AngleWeightedReductionFeatureAggregator)
    mod("pytorch3d.implicitron.models.view_pooler")
    mod("pytorch3d.implicitron.models.view_pooler.feature_aggregator", FeatureAggregatorBase=FeatureAggregatorBase,
        AngleWeightedReductionFeatureAggregator=AngleWeightedReductionFeatureAggregator)
    sys.modules.update(mods)
    return registry
