"""Minimal stand-in for the PyTorch3D Implicitron registry/config surface.

The reference plugs its denoiser / implicit function / renderer / model into
``pytorch3d.implicitron.tools.config`` (``@registry.register``, ``registry.get(Base, name)``,
``ReplaceableBase``, ``Configurable``, ``<member>_class_type`` + ``<member>_<Type>_args``;
e.g. /root/reference/holo_diffusion/utils/diffusion_utils.py:41-42,
holo_diffusion_model.py:44-55,118-130).  PyTorch3D is not installed on the build or GPU boxes,
so this module provides the same names with the same lookup semantics for the five plugin
classes of the hot path.  When PyTorch3D *is* importable, :func:`register_with_pytorch3d` also
registers the classes in the real registry.
"""
from __future__ import annotations

import dataclasses
import enum
from typing import Any, Dict, Type


class Configurable:
    """Marker base: classes whose public annotated fields are configuration (dataclass-like)."""

    @classmethod
    def config_fields(cls) -> Dict[str, Any]:
        out = {}
        for klass in reversed(cls.__mro__):
            if not (isinstance(klass, type) and issubclass(klass, Configurable)):
                continue
            for k in klass.__dict__.get("__annotations__", {}):
                if not k.startswith("_") and hasattr(klass, k):
                    out[k] = getattr(klass, k)
        return out


class ReplaceableBase(Configurable):
    """Marker base for pluggable implementations selected by ``<member>_class_type``."""


class _Registry:
    def __init__(self):
        self._by_base: Dict[Type, Dict[str, Type]] = {}

    def register(self, cls: Type) -> Type:
        bases = [b for b in cls.__mro__[1:] if isinstance(b, type) and issubclass(b, ReplaceableBase)
                 and b is not ReplaceableBase]
        if not bases:
            raise ValueError(f"{cls.__name__} does not derive from a ReplaceableBase subclass")
        for b in bases:
            self._by_base.setdefault(b, {})[cls.__name__] = cls
        return cls

    def get(self, base: Type, name: str) -> Type:
        try:
            return self._by_base[base][name]
        except KeyError:
            raise ValueError(f"{name} has not been registered as a {base.__name__}") from None

    def get_all(self, base: Type):
        return list(self._by_base.get(base, {}).values())


registry = _Registry()


def get_default_args(cls: Type) -> Dict[str, Any]:
    """Default config of a Configurable class as a plain dict (OmegaConf-free)."""
    out = {}
    for k, v in cls.config_fields().items():
        if dataclasses.is_dataclass(v):
            v = dataclasses.asdict(v)
        out[k] = list(v) if isinstance(v, tuple) else v
    return out


def apply_config(obj: Any, kwargs: Dict[str, Any]) -> None:
    """Set config fields from kwargs, rejecting unknown keys like the Implicitron dataclasses do."""
    fields = type(obj).config_fields()
    for k, v in kwargs.items():
        if k not in fields:
            raise TypeError(f"{type(obj).__name__} got an unexpected config field '{k}'")
        default = fields[k]
        if isinstance(default, tuple) and isinstance(v, (list, tuple)):
            v = tuple(v)
        if isinstance(default, enum.Enum) and isinstance(v, str):  # YAML carries enum fields by member name
            v = type(default)[v]
        setattr(obj, k, v)
    for k, v in fields.items():
        if k not in kwargs:
            setattr(obj, k, v)


def register_with_pytorch3d(*classes: Type) -> bool:
    """Best-effort registration in the real Implicitron registry (no-op when PyTorch3D is absent)."""
    try:
        from pytorch3d.implicitron.tools.config import registry as p3d_registry  # type: ignore
    except Exception:
        return False
    for c in classes:
        try:
            p3d_registry.register(c)
        except Exception:
            pass
    return True
